"""Host-side helpers of the JRDB loader (the reference imports them as ``data.utils.utils`` and
``data.utils.jrdb_transforms``, dataloader.py:8-9; neither module is in the reference repository)."""
