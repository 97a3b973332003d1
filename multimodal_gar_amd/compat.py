"""Import aliases for the reference's caller code.

``train_func.py`` does ``from model.gat_model import *`` / ``from pcdet.config import cfg, ...`` /
``from dataloader import JRDB_act`` (train_func.py:20-34) and ``dataloader.py`` imports ``data.utils.*``
(dataloader.py:8-9).  ``install()`` registers this package's ``model``, ``pcdet`` and ``data``
sub-packages and its ``dataloader`` module under those top-level names so such a script runs
unchanged on top of the MI355X operator set:

    import multimodal_gar_amd.compat as compat; compat.install()
    from model.gat_model import GAR_Fusion_ALL
    from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils
    from dataloader import JRDB_act
"""
import importlib
import importlib.util
import sys


TOPS = ("model", "pcdet", "data", "dataloader")


def install(force=False):
    for top in TOPS:
        if top in sys.modules and not force:
            mod = sys.modules[top]
            if not getattr(mod, "__name__", "").startswith("multimodal_gar_amd"):
                raise ImportError("a different top-level %r package is already imported" % top)
            continue
        pkg = importlib.import_module("multimodal_gar_amd." + top)
        sys.modules[top] = pkg
        prefix = "multimodal_gar_amd." + top + "."
        for name, mod in list(sys.modules.items()):
            if name.startswith(prefix):
                sys.modules[top + "." + name[len(prefix):]] = mod
    sys.meta_path.insert(0, _AliasFinder())


class _AliasFinder:
    """Resolves not-yet-imported ``model.x`` / ``pcdet.x.y`` names to the in-package modules."""

    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top not in TOPS or fullname in sys.modules:
            return None
        real = "multimodal_gar_amd." + fullname
        try:
            mod = importlib.import_module(real)
        except ImportError:
            return None
        # NOT registered in sys.modules here: importlib's _find_spec would then discard this spec in favour of the
        # module's own __spec__ and execute the source file a second time; _Preloaded.create_module hands the
        # already-imported module over and the import system registers it under the alias
        return importlib.util.spec_from_loader(fullname, _Preloaded(mod))


class _Preloaded:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass
