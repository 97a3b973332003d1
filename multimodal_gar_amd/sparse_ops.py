"""Sparse 3-D convolution on the kernels of csrc/sparse_conv.hip: voxel hash table, rulebook (neighbour tables built on
the device) and the gather-GEMM convolution with its two gradients.  This is what stands behind the ``spconv`` namespace
of pcdet/utils/spconv_utils.py (the reference imports the third-party spconv 2.2.3 there: spconv_utils.py:3-6).

Semantics (published definition of the two layers; spconv itself is neither vendored nor installed):
  out[o] = sum_k W_k . in[i]   over the kernel offsets k whose input site i = o * stride - pad + k is active;
  SubMConv3d   : output sites = input sites (stride 1, "same" padding);
  SparseConv3d : output sites = every o in the output grid with at least one active input under its kernel, listed in
                 ascending (b, z, y, x) order; output grid = floor((in + 2 pad - k) / stride) + 1.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L

INT_MAX = 2 ** 31 - 1


def _triple(v):
    return tuple(int(x) for x in v) if isinstance(v, (list, tuple)) else (int(v),) * 3


class VoxelHash:
    """(b, z, y, x) -> row id over the active voxels of a (Z, Y, X) grid: the sparse replacement of the dense (B, Z, Y, X)
    table of generate_voxel2pinds (reference pcdet/utils/common_utils.py:244-252; 80 MB per sample at the shipped grid)."""

    def __init__(self, indices, spatial_shape):
        assert indices.is_cuda and indices.dtype == torch.int32 and indices.dim() == 2 and indices.shape[1] == 4
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        n = indices.shape[0]
        cap = 16
        while cap < 2 * n:
            cap *= 2
        self.capacity = cap
        self.keys = torch.full((cap,), -1, dtype=torch.int64, device=indices.device)
        self.vals = torch.full((cap,), INT_MAX, dtype=torch.int32, device=indices.device)
        z, y, x = self.spatial_shape
        L.call("mgar_voxel_hash_build", n, L.iptr(self.indices), z, y, x, L.dev_ptr(self.keys, torch.int64), L.iptr(self.vals), cap,
               L.stream_of(indices))

    def lookup(self, coords):
        """coords (M, 4) int32 [b, z, y, x] -> rows (M) int32, -1 where no active voxel."""
        coords = coords.contiguous().int()
        rows = torch.empty((coords.shape[0],), dtype=torch.int32, device=coords.device)
        z, y, x = self.spatial_shape
        L.call("mgar_voxel_hash_lookup", coords.shape[0], L.iptr(coords), z, y, x, L.dev_ptr(self.keys, torch.int64), L.iptr(self.vals),
               self.capacity, L.iptr(rows), L.stream_of(coords))
        return rows


class Rulebook:
    """Neighbour tables of one (input sites, kernel, stride, padding) combination; shared by every convolution with the
    same ``indice_key`` (as spconv shares its indice pairs).

    Precondition (spconv's as well): the rows of ``indices`` are UNIQUE voxel coordinates -- what a voxeliser produces
    (csrc/voxel_hash.hpp keeps the smallest row of a duplicated coordinate, so a duplicate would simply never be read; but the
    pair-list data gradient, csrc/sparse_conv.hip spconv_pairs_gemm_kernel, accumulates per input row without atomics and relies on
    every (output, offset) pair naming one row).  ``check_unique=True`` asserts it with one sort (off by default: a host sync)."""

    def __init__(self, indices, spatial_shape, batch_size, kernel, stride, padding, subm, check_unique=False):
        import ctypes
        if check_unique and indices.shape[0] > 1:
            sz, sy, sx = (int(v) for v in spatial_shape)
            key = ((indices[:, 0].long() * sz + indices[:, 1].long()) * sy + indices[:, 2].long()) * sx + indices[:, 3].long()
            if torch.unique(key).numel() != key.numel():
                raise ValueError("sparse convolution: duplicate voxel coordinates in the input indices")
        self.kernel, self.stride, self.padding, self.subm = _triple(kernel), _triple(stride), _triple(padding), bool(subm)
        kz, ky, kx = self.kernel
        self.K = kz * ky * kx
        self.in_indices = indices.contiguous()
        self.in_shape = [int(s) for s in spatial_shape]
        dev = indices.device
        in_hash = VoxelHash(self.in_indices, self.in_shape)
        if self.subm:
            assert self.stride == (1, 1, 1) and all(2 * p == k - 1 for p, k in zip(self.padding, self.kernel)), \
                "submanifold convolution: stride 1, odd kernel, padding k // 2"
            self.out_indices, self.out_shape = self.in_indices, list(self.in_shape)
        else:
            self.out_shape = [(s + 2 * p - k) // st + 1 for s, p, k, st in zip(self.in_shape, self.padding, self.kernel, self.stride)]
        geom = (ctypes.c_int * 15)(*self.kernel, *self.stride, *self.padding, *self.in_shape, *self.out_shape)
        if not self.subm:
            self.out_indices = self._output_sites(geom)
        n_out = self.out_indices.shape[0]
        self.nbr = torch.empty((n_out, self.K), dtype=torch.int32, device=dev)
        L.call("mgar_spconv_rulebook", n_out, L.iptr(self.out_indices), geom, L.dev_ptr(in_hash.keys, torch.int64), L.iptr(in_hash.vals),
               in_hash.capacity, 0, L.iptr(self.nbr), L.stream_of(indices))
        self._pairs = None
        self._geom = geom
        self.inv = None             # built on demand (inverse_table): only the table-driven data gradient of a strided conv reads it

    def pair_count(self):
        """Number of (output site, offset) pairs with a neighbour (host int; one sync, cached: instrumentation only)."""
        if getattr(self, "_pair_count", None) is None:
            self._pair_count = int((self.nbr >= 0).sum().item())
        return self._pair_count

    def inverse_table(self):
        """(N_in, K) int32: the output row input i reaches through offset k, or -1 (strided convolutions; a submanifold one uses
        its forward table with mirrored offsets)."""
        if self.inv is None and not self.subm:
            dev = self.in_indices.device
            out_hash = VoxelHash(self.out_indices, self.out_shape)
            self.inv = torch.empty((self.in_indices.shape[0], self.K), dtype=torch.int32, device=dev)
            L.call("mgar_spconv_rulebook", self.in_indices.shape[0], L.iptr(self.in_indices), self._geom,
                   L.dev_ptr(out_hash.keys, torch.int64), L.iptr(out_hash.vals), out_hash.capacity, 1, L.iptr(self.inv),
                   L.stream_of(self.in_indices))
        return self.inv

    def pairs(self):
        """The neighbour table compacted per kernel offset (built once per rulebook, on first use):
        pair_i / pair_o (P) int32 -- input / output row of every (offset, output site) with a neighbour, grouped by offset,
        ascending output row inside an offset; items (n_items, 4) int32 {offset, first pair, end pair, 0} of at most
        mgar_spconv_pair_chunk() pairs, item_start (K + 1) int32 and the item count for the weight gradient; a finer cut of the
        same pairs and its item_start as a host array for the per-offset launches of the forward / data gradient."""
        if self._pairs is None:
            dev = self.nbr.device
            n_out = self.nbr.shape[0]
            # csrc/sparse_conv.hip, sp_pairs_*: count per (offset, row block) -> scan -> stable fill; three launches
            nblk = max(L.raw("mgar_spconv_pairs_blocks", n_out), 1)
            blk = torch.empty((self.K, nblk), dtype=torch.int32, device=dev)
            total = torch.empty((self.K,), dtype=torch.int32, device=dev)
            st = L.stream_of(self.nbr)
            L.call("mgar_spconv_pairs_count", n_out, self.K, L.iptr(self.nbr), L.iptr(blk), L.iptr(total), st)
            counts = total.tolist()                                              # one host synchronisation per rulebook
            self._pair_count = sum(counts)
            starts = torch.tensor([0] + counts[:-1], dtype=torch.int64).cumsum(0).to(dev)
            pair_i = torch.empty((max(self._pair_count, 1),), dtype=torch.int32, device=dev)
            pair_o = torch.empty_like(pair_i)
            L.call("mgar_spconv_pairs_fill", n_out, self.K, L.iptr(self.nbr), L.iptr(blk), L.dev_ptr(starts, torch.int64), L.iptr(pair_i),
                   L.iptr(pair_o), st)
            chunk = L.raw("mgar_spconv_pair_chunk")

            def cut(chunk_of):
                import numpy as np
                rows, item_start, at = [], [0], 0
                for k, n in enumerate(counts):
                    if n:
                        c = chunk_of(n)
                        b = np.arange(0, n, c, dtype=np.int64)
                        rows.append(np.stack([np.full_like(b, k), at + b, at + np.minimum(b + c, n), np.zeros_like(b)], 1))
                    at += n
                    item_start.append(item_start[-1] + (len(rows[-1]) if n else 0))
                arr = np.concatenate(rows, 0).astype(np.int32) if rows else np.zeros((1, 4), np.int32)
                return torch.from_numpy(arr).to(dev), item_start, item_start[-1]
            # weight gradient: all offsets in one launch, long items (few partials).  forward / data gradient: one launch per
            # offset, so an offset's pairs are cut finely enough to fill the chip (~1 024 workgroups per launch)
            items_dw, start_dw, n_dw = cut(lambda n: chunk)
            items_fw, start_fw, _ = cut(lambda n: min(chunk, max(64, -(-n // (1024 * 64)) * 64)))
            import ctypes
            self._pairs = (pair_i, pair_o, items_dw, torch.tensor(start_dw, dtype=torch.int32, device=dev), n_dw,
                           items_fw, (ctypes.c_int * len(start_fw))(*start_fw))
        return self._pairs

    def _output_sites(self, geom):
        """Active output sites of a strided convolution: every o = (i + pad - k) / stride that is integral and inside the
        output grid, over all active inputs i and offsets k (one kernel), made unique: ascending in (b, z, y, x)."""
        n_in = self.in_indices.shape[0]
        keys = torch.empty((n_in, self.K), dtype=torch.int64, device=self.in_indices.device)
        L.call("mgar_spconv_output_keys", n_in, L.iptr(self.in_indices), geom, L.dev_ptr(keys, torch.int64), L.stream_of(keys))
        keys = keys.view(-1)
        uniq = torch.unique(keys[keys >= 0])
        zo, yo, xo = self.out_shape
        b = uniq // (zo * yo * xo)
        r = uniq % (zo * yo * xo)
        return torch.stack([b, r // (yo * xo), (r % (yo * xo)) // xo, r % xo], 1).int().contiguous()


def _pow2(v):
    return v >= 1 and (v & (v - 1)) == 0


def _note(kernel, rb, c_src, c_dst, n_src, n_dst, rmw):
    """Instrumented runs (bench.py): credit a sparse product's algorithmic work -- 2 P C_src C_dst flop over the P (site,
    offset) pairs of the rulebook; bytes = source rows once + destination rows once (twice for the read-modify-write pair
    kernels) + 8 B of pair indices each."""
    if not L.kernel_timing_on():
        return
    p = rb.pair_count()
    L.note_work(kernel, flops=2.0 * p * c_src * c_dst,
                nbytes=4.0 * (n_src * c_src + (2 if rmw else 1) * n_dst * c_dst) + 8.0 * p)


def _gather_gemm(n_out, k, cin, cout, feats, nbr, w, flip):
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    L.call("mgar_spconv_gather_gemm", n_out, k, cin, cout, L.fptr(feats), L.iptr(nbr), L.fptr(w), int(flip), L.fptr(out),
           L.stream_of(feats))
    return out


def _pairs_gemm(rb, n_dst, src_feats, w, transpose):
    """dst (n_dst, Cd) = sum_k over the pairs of offset k of src[.] . w[k] (csrc/sparse_conv.hip, spconv_pairs_gemm_kernel).
    transpose=False: forward (input rows -> output rows, w (K, Cin, Cout)); True: data gradient (dout rows -> input rows, w
    (K, Cout, Cin) = W_k^T)."""
    import ctypes
    pair_i, pair_o, _, _, _, items, start_host = rb.pairs()
    k, cs, cd = w.shape
    dst = torch.zeros((n_dst, cd), dtype=torch.float32, device=src_feats.device)
    ps, pd = (pair_o, pair_i) if transpose else (pair_i, pair_o)
    L.call("mgar_spconv_pairs_gemm", k, cs, cd, L.fptr(src_feats), L.iptr(ps), L.iptr(pd), L.iptr(items),
           ctypes.cast(start_host, ctypes.c_void_p), L.fptr(w), L.fptr(dst), L.stream_of(src_feats))
    return dst


PAIRS_FORWARD = False    # forward over pair lists (False: the table-driven, output-stationary gather-GEMM -- no read-modify-write)
# data gradient: over pair lists (27 per-offset launches with read-modify-write, no inverse table), or table-driven on the
# register-gather kernel (one launch, no read-modify-write; a strided convolution needs its inverse table: one more rulebook
# launch).  MGAR_SPCONV_DGRAD=pairs|table overrides (A/B in profiles/).
PAIRS_DGRAD = os.environ.get("MGAR_SPCONV_DGRAD", "table") == "pairs"


class _SparseConv(Function):
    """out (No, Cout) = sum_k feats[nbr[:, k]] @ w[k]  with w (K, Cin, Cout)."""

    @staticmethod
    def forward(ctx, feats, w, rb):
        feats, w = feats.contiguous().float(), w.contiguous().float()
        k, cin, cout = w.shape
        if PAIRS_FORWARD and _pow2(cin) and max(cin, cout) <= 128:
            out = _pairs_gemm(rb, rb.nbr.shape[0], feats, w, False)
            _note("spconv_gemm", rb, cin, cout, feats.shape[0], out.shape[0], True)
        else:
            out = _gather_gemm(rb.nbr.shape[0], k, cin, cout, feats, rb.nbr, w, 0)
            _note("spconv_gemm", rb, cin, cout, feats.shape[0], 0, False)       # the output write is in the launcher's figure
        ctx.save_for_backward(feats, w)
        ctx.rb = rb
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        feats, w = ctx.saved_tensors
        rb = ctx.rb
        k, cin, cout = w.shape
        dout = dout.contiguous().float()
        dfeats = dw = None
        if ctx.needs_input_grad[0]:
            wt = w.transpose(1, 2).contiguous()                                   # (K, Cout, Cin)
            if PAIRS_DGRAD and _pow2(cout) and max(cin, cout) <= 128:
                dfeats = _pairs_gemm(rb, feats.shape[0], dout, wt, True)
                _note("spconv_gemm", rb, cout, cin, dout.shape[0], feats.shape[0], True)
            else:
                dfeats = _gather_gemm(feats.shape[0], k, cout, cin, dout, rb.nbr if rb.subm else rb.inverse_table(), wt, 1 if rb.subm else 0)
        if ctx.needs_input_grad[1] and _pow2(cin) and _pow2(cout) and max(cin, cout) <= 128:
            # pair lists: no work on sites without a neighbour under the offset, tiles software-pipelined (csrc/sparse_conv.hip)
            pair_i, pair_o, items, item_start, n_items = rb.pairs()[:5]
            part = torch.empty((max(n_items, 1), cin, cout), dtype=torch.float32, device=feats.device)
            dw = torch.empty((k, cin, cout), dtype=torch.float32, device=feats.device)
            L.call("mgar_spconv_pairs_dw", n_items, k, cin, cout, L.fptr(feats), L.fptr(dout), L.iptr(pair_i), L.iptr(pair_o), L.iptr(items),
                   L.iptr(item_start), L.fptr(part), L.fptr(dw), L.stream_of(feats))
            _note("spconv_dw", rb, cin, cout, feats.shape[0], dout.shape[0], False)
        elif ctx.needs_input_grad[1]:
            n_out = rb.nbr.shape[0]
            nchunk = L.raw("mgar_spconv_dw_chunks", n_out)
            part = torch.empty((max(nchunk, 1), k, cin, cout), dtype=torch.float32, device=feats.device)
            L.call("mgar_spconv_dw", n_out, k, cin, cout, L.fptr(feats), L.iptr(rb.nbr), L.fptr(dout), L.fptr(part), L.stream_of(feats))
            dw = part.sum(0) if nchunk > 0 else torch.zeros_like(w)
        return dfeats, dw, None


def sparse_conv3d(features, indices, spatial_shape, batch_size, weight, kernel, stride, padding, subm, cache, key):
    """One sparse convolution.  weight: the module's parameter in spconv 2.x layout (Cout, kz, ky, kx, Cin).
    -> (out_features (No, Cout), out_indices (No, 4) int32, out_spatial_shape).  ``cache`` / ``key``: the rulebook of an
    indice_key is built once per sparse tensor lineage and shared (spconv's indice_dict)."""
    rb = cache.get(key) if key is not None else None
    if rb is not None and rb.in_indices.data_ptr() == indices.data_ptr() and rb.in_indices.shape == indices.shape:
        # same lineage: the geometry must be the one the table was built for (spconv asserts the same on a shared indice_key)
        geometry = (_triple(kernel), _triple(stride), _triple(padding), bool(subm), [int(s) for s in spatial_shape])
        if geometry != (rb.kernel, rb.stride, rb.padding, rb.subm, rb.in_shape):
            raise ValueError("indice_key %r is shared by convolutions of different geometry: %r vs %r"
                             % (key, geometry, (rb.kernel, rb.stride, rb.padding, rb.subm, rb.in_shape)))
    else:
        rb = Rulebook(indices, spatial_shape, batch_size, kernel, stride, padding, subm)
        if key is not None:
            cache[key] = rb
    cout, cin = weight.shape[0], weight.shape[-1]
    w = weight.permute(1, 2, 3, 4, 0).reshape(rb.K, cin, cout)
    out = _SparseConv.apply(features, w, rb)
    return out, rb.out_indices, rb.out_shape
