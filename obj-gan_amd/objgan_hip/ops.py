"""torch.autograd bindings of the gfx950 kernels (host side of the C-ABI in include/objgan_hip.h).

PyTorch is used here for device memory, streams and the autograd tape only: every
operator below launches hand-written HIP kernels through ctypes.  There is no CPU or
eager-PyTorch fallback -- a CPU tensor (or a missing libobjgan_hip.so) raises.
"""
import ctypes
import math

import os

import torch

from . import _lib

_F32 = torch.float32


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    # the raw handle of torch's current stream (torch.cuda.current_stream() builds a Stream object: 9 us per call,
    # 7 ms of host time per training step)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.ObjganHipError(
                "objgan_hip operators run on the MI355X only (got a %s tensor); there is no CPU path"
                % t.device)
        if t.dtype != _F32:
            raise _lib.ObjganHipError("objgan_hip operators are fp32 (got %s)" % t.dtype)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_IARR = {}


def _iarr(vals):
    """ctypes int array of a small index list; the arrays of the tap tables are built once per distinct list (the host
    issues ~450 convolutions per training step, three lists each)"""
    key = tuple(vals)
    a = _IARR.get(key)
    if a is None:
        if len(_IARR) > 4096:
            _IARR.clear()
        a = _IARR[key] = (ctypes.c_int * len(key))(*[int(v) for v in key])
    return a


_QUERY = {}


def _q(name, *args):
    """memoized host-only query of the library (launch plans, workspace sizes, bank layouts: pure functions of their integer
    arguments; the development build's environment knobs are read once per process anyway)"""
    key = (name,) + args
    v = _QUERY.get(key)
    if v is None:
        if len(_QUERY) > 65536:
            _QUERY.clear()
        v = _QUERY[key] = getattr(_lib.load(), name)(*args)
    return v


# ----------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------
_ACT = {None: 0, "none": 0, "lrelu": 1, "tanh": 2, "sigmoid": 3, "relu": 4}

# Arithmetic of the MFMA convolution kernels:
#   "fp32"    fp32 operands on v_mfma_f32_32x32x2_f32 (the matrix pipe's native fp32 rate: 157 TFLOP/s)
#   "bf16x3"  fp32 operands, each split EXACTLY into three bf16 pieces (24 = 8 + 8 + 8 significand bits), the six
#             largest of the nine partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32 results
#             (the dropped products are below 2^-24 of a product, 30x below the accumulation rounding both modes
#             share) at 2.67x the native fp32 matrix rate -- csrc/conv_igemm.hip `og_split8`; the same scheme as
#             oneMKL's BF16x3 / cuBLAS's BF16x9 fp32 emulation
#   "fp16x2"  fp32 operands, each as TWO fp16 pieces of x * 2^s (s from the tensor's maximum: |x * 2^s - h - l| <= 2^-23 |x|
#             for every element within 2^-10 of the maximum), the three products hh, hl, lh on v_mfma_f32_32x32x16_f16
#             with fp32 accumulation, the scales undone exactly in the epilogue: fp32 results at HALF the matrix work of
#             bf16x3 -- measured against fp64 its error is below bf16x3's and the fp32 MFMA's.  Taken by the large
#             launches (>= _H2_MIN_FLOP: each needs a maximum pass over its pixel operand); the small ones, the
#             LDS-staged weight-gradient kernels and the thin VALU kernels keep their arithmetic (bf16x3 / fp32).
#   "bf16"    mixed precision (BASELINE config 5): operands ROUNDED to bf16 at the matrix-core inputs, fp32
#             accumulation, fp32 tensors / master weights / norm statistics
import os as _os
_MATH_IDS = {"fp32": 0, "bf16": 1, "bf16x3": 2, "fp16x2": 4}
# default: fp16x2 (round 4; rounds 1-2: fp32, round 3: bf16x3) -- fp32 results; measured against fp64 the error of both
# split arithmetics is at or below the fp32 MFMA's on every operator (tests/test_kernels_gpu.py, profiles/r04_parity.txt)
if _os.environ.get("OBJGAN_CONV_MATH", "fp16x2") not in _MATH_IDS:      # a typo must not silently change the arithmetic
    raise _lib.ObjganHipError("OBJGAN_CONV_MATH=%r: must be one of %s"
                              % (_os.environ["OBJGAN_CONV_MATH"], sorted(_MATH_IDS)))
_MATH = {"mode": _MATH_IDS[_os.environ.get("OBJGAN_CONV_MATH", "fp16x2")]}


def set_conv_math(mode):
    if mode not in _MATH_IDS:
        raise _lib.ObjganHipError("conv math must be one of %s" % (sorted(_MATH_IDS),))
    _MATH["mode"] = _MATH_IDS[mode]


def get_conv_math():
    return [k for k, v in _MATH_IDS.items() if v == _MATH["mode"]][0]


# fp16x2: launches below this keep bf16x3 (the maximum pass would cost more than it saves)
_H2_MIN_FLOP = float(_os.environ.get("OBJGAN_H2_MIN_FLOP", "1.0e9"))


def _call_math(flop):
    """arithmetic id of ONE matrix-kernel call under the current mode"""
    m = _MATH["mode"]
    if m == 4 and flop < _H2_MIN_FLOP:
        return 2
    return m


_AMAX_SLOTS = 1024          # csrc/common.h OG_AMAX_SLOTS


def _amax_wanted(numel):
    """does a producer of a tensor of this size emit its partial maxima?  (fp16x2 mode; tensors a launch above the
    FLOP threshold could read: anything of a few hundred thousand elements up)"""
    return _MATH["mode"] == 4 and numel >= 262144


_ZERO_ROWS = {}             # HIP stream -> [chunk of zeroed slot rows, rows handed out]
_ZERO_CHUNK = 32


_CAPTURE = {"on": False}


def capture_begin():
    """A hipGraph capture of a chain of these operators starts (objgan_hip.graphs): zeroed slot rows handed out from now
    on must come from a chunk whose fill launch is INSIDE the capture (a replay re-zeroes it), never from a chunk an eager
    launch filled earlier on a stream the capture happens to reuse."""
    _ZERO_ROWS.clear()
    _CAPTURE["on"] = True


def capture_end():
    _ZERO_ROWS.clear()          # the chunks of the capture belong to its replays: eager producers start a fresh one
    _CAPTURE["on"] = False


def _amax_zeroed(device):
    """1024 zeroed maximum slots for a producer that adds into them atomically: a row of a chunk zero-filled by ONE
    launch per 32 rows (it was one fill launch per producer call).  Per HIP stream: the fill and the producers that use
    its rows are ordered on the stream that asked; a row is handed out once and lives as long as its tensor does."""
    sid = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    ent = _ZERO_ROWS.get(sid)
    if ent is None or ent[1] >= _ZERO_CHUNK or ent[0].device != device:
        ent = _ZERO_ROWS[sid] = [torch.zeros((_ZERO_CHUNK, _AMAX_SLOTS), dtype=_F32, device=device), 0]
    row = ent[0][ent[1]]
    ent[1] += 1
    return row


def _amax_attach(t, amax):
    """the producer of `t` filled `amax` (its partial maxima) on the current stream: what _absmax(t) returns from now on"""
    sid = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    try:
        t._og_absmax = {sid: (t._version, amax)}
    except (AttributeError, RuntimeError):
        pass


_ABSMAX_LOG = {} if _os.environ.get("OBJGAN_H2_LOG") == "1" else None
if _ABSMAX_LOG is not None:
    import atexit

    def _dump_absmax_log():
        import sys
        for (shape, where), n in sorted(_ABSMAX_LOG.items(), key=lambda kv: -kv[1] * max(1, int(torch.tensor(kv[0][0]).prod()))):
            sys.stderr.write("ABSMAX %6d x %-24s %s\n" % (n, shape, where))
    atexit.register(_dump_absmax_log)


def _absmax(t):
    """the 1024 partial maxima of |t| (fp16x2's scale input): one pass over t, no host sync.  Cached on the tensor
    object for the duration of its life (a forward activation is the pixel operand of its convolution AND, in the
    backward pass, the column operand of that layer's weight gradient) -- per HIP stream: a tensor shared by jobs on
    different streams (the images every discriminator reads) gets one pass per stream, ordered with that stream's
    kernels."""
    sid = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    cache = getattr(t, "_og_absmax", None)
    if cache is not None:
        got = cache.get(sid)
        if got is not None and got[0] == t._version:
            if _H2_VERIFY["on"]:
                _verify_amax(t, got[1])
            return got[1]
    if _ABSMAX_LOG is not None:          # (development: who still needs a pass?)
        import traceback
        key = (tuple(t.shape), " < ".join("%s:%d" % (f.name, f.lineno) for f in traceback.extract_stack(limit=7)[:-1][::-1][:5]))
        _ABSMAX_LOG[key] = _ABSMAX_LOG.get(key, 0) + 1
    out = torch.empty(_AMAX_SLOTS, dtype=_F32, device=t.device)
    src = t if (t.is_contiguous() and not (t.data_ptr() & 15)) else t.contiguous().clone()
    _lib.call("objgan_absmax_partials", _p(src), src.numel(), _p(out), _stream())
    try:
        if cache is None:
            cache = t._og_absmax = {}
        cache[sid] = (t._version, out)
    except (AttributeError, RuntimeError):
        pass
    return out


# fp16x2 on records (round 5): the MFMA implicit-GEMM kernel reads its pixel operand as PRE-SPLIT fp16 pieces,
# rec[n][c / 16][h | l][pixel][c % 16] with x * 2^s = h + l (csrc/conv_igemm_rec.hip), instead of gathering the fp32 NCHW
# tensor and splitting it inside the loop -- same products in the same order, bit-identical results.  The record of a
# tensor is a pure function of (tensor, partial maxima); it is written by one pass (objgan_h2_records) on first use and
# kept on the tensor object like the maxima (per HIP stream, invalidated by the version counter).
_REC = {"on": _os.environ.get("OBJGAN_H2_RECORDS", "1") != "0",
        # A record written by its own pass costs 8 bytes of HBM traffic per element of the tensor; the kernel that reads it
        # gains ~15 % (tall 8-wave tiles) to ~30-55 % (block rows <= 96 rows with two pixel groups per wave).  The pass pays
        # when the launch does enough arithmetic per element of its pixel operand: 2 * M * taps * (output / input pixels)
        # flop per element (profiles/r05_records_convbench.txt).  A tensor that already carries its record is always read
        # through it.
        # the weight gradient reads x through its record where that pays ("all": wherever the geometry allows; "0": never)
        "wgrad": {"0": False, "all": "all"}.get(_os.environ.get("OBJGAN_REC_WGRAD", "1"), True),
        # 6: the record-form weight gradient reads dy pre-split too (its fp16 pair, one pass into the call's workspace).
        # Built and measured in round 6: bit-identical, and NOT faster -- the pair pass costs more than the in-loop split of
        # dy it removes (every eligible weight gradient on records: 137.0 ms per step with the in-loop split, 138.6 with the
        # pair, profiles/r06_ab_variants.txt): the record kernel is bound by its LDS fragment reads, not by the VALU.  Off.
        # 7: two column groups per wave (conv_wgrad_rec2_kernel)
        "wgrad_math": int(_os.environ.get("OBJGAN_REC_WGRAD_MATH", "6" if _os.environ.get("OBJGAN_REC_WGRAD_DYP", "0") == "1" else "5")),
        "min_i": float(_os.environ.get("OBJGAN_REC_MIN_I", "2500")),
        "min_i_short": float(_os.environ.get("OBJGAN_REC_MIN_I_SHORT", "1200"))}


def set_h2_records(on):
    _REC["on"] = bool(on)


def _records(t, amax, N, C, HW, intensity=None, rows=0):
    """the fp16x2 record of the contiguous fp32 tensor t [N, C, HW] under the scale of `amax` (None: the launch that
    asks does `intensity` flop per element of t with `rows` output rows -- not enough to pay for the pass)"""
    sid = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    cache = getattr(t, "_og_rec", None)
    if cache is not None:
        got = cache.get(sid)
        if got is not None and got[0] == t._version and got[2] is amax:
            return got[1]
    if intensity is not None and intensity < (_REC["min_i_short"] if rows <= 96 else _REC["min_i"]):
        return None
    if N > 65535 or float(N) * ((C + 15) // 16 * 16) * HW * 4.0 >= 4.0e9:
        return None                 # beyond objgan_h2_records' 32-bit offsets: the launch keeps the gather form (math 4)
    rec = torch.empty(N * ((C + 15) // 16 * 16) * HW, dtype=_F32, device=t.device)
    _lib.call("objgan_h2_records", _p(t), _p(amax), _p(rec), N, C, HW, _stream())
    try:
        if cache is None:
            cache = t._og_rec = {}
        cache[sid] = (t._version, rec, amax)
    except (AttributeError, RuntimeError):
        pass
    return rec


# bf16 mode (BASELINE config 5): the matrix kernels read their pixel operand from a bf16 channel-blocked copy.  Round 6: the
# copy of a tensor is made ONCE (objgan_nhwc_bf16) and kept on the tensor object like its record -- the forward convolution
# and the weight gradient of a layer, and every branch of an Inception block, read the same copy (math 3) instead of
# writing their own into their workspace on every call (math 1: 32.9 of 216 ms per B = 32 step, 1 161 copy launches).
_B16 = {"on": _os.environ.get("OBJGAN_BF16_COPY_CACHE", "1") != "0"}


def _bf16_copy(t, N, C, HW):
    """the bf16 channel-blocked copy of the contiguous fp32 tensor t [N, C, HW] (None: too large -- the call keeps math 1)"""
    sid = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    cache = getattr(t, "_og_b16", None)
    if cache is not None:
        got = cache.get(sid)
        if got is not None and got[0] == t._version:
            return got[1]
    nfl = _q("objgan_nhwc_bf16_floats", N, C, HW)
    if nfl <= 0 or (t.data_ptr() & 15):
        return None
    buf = torch.empty(nfl, dtype=_F32, device=t.device)
    _lib.call("objgan_nhwc_bf16", _p(t), _p(buf), N, C, HW, _stream())
    try:
        if cache is None:
            cache = t._og_b16 = {}
        cache[sid] = (t._version, buf)
    except (AttributeError, RuntimeError):
        pass
    return buf


# Packed filter banks are kept while their weights are unchanged.  A weight tensor is eligible
# when something vouches for its contents: either it carries `_og_epoch` (a one-element list owned
# by its optimizer arena, bumped by every ArenaAdam.step -- the fused Adam kernel writes through a
# raw pointer, invisible to torch's version counter), or it is frozen (requires_grad False, e.g.
# the Inception encoder); torch-side in-place edits are caught by `_version`.  The C library stays
# stateless: the cache is host-side memory ownership, exactly like any other scratch buffer.
#
# One entry per (weight, direction, taps, bank layout, math): the entry records the version / epoch its
# bank was packed for; a stale bank is re-packed IN PLACE (no new buffer), either by the conv call that
# finds it stale or -- for the banks of an optimizer arena -- all at once by repack_arena() right behind
# the Adam kernel, from a device table of pack jobs (one launch instead of ~60 per network and step).
class _Bank(object):
    __slots__ = ("w", "wt", "version", "epoch", "jobs")

    def __init__(self, w, wt):
        self.w, self.wt, self.version, self.epoch, self.jobs = w, wt, None, None, None


_PACK_CACHE = {}
_PACK_CACHE_MAX = 4096
_ARENA_BANKS = {}       # id(epoch cell) -> {"cell": cell, "banks": [...], "table": device bytes or None, "n": jobs}


def invalidate_packed():
    _PACK_CACHE.clear()
    _ARENA_BANKS.clear()


def _pack_key(w, transpose, src_tap, big, math=None):
    ep = getattr(w, "_og_epoch", None)
    if (ep is None and w.requires_grad) or getattr(w, "_og_nocache", False):
        return None
    return (w.data_ptr(), tuple(w.shape), int(transpose), tuple(src_tap), big, _MATH["mode"] if math is None else math)


def _epoch_of(w):
    ep = getattr(w, "_og_epoch", None)
    return ep[0] if ep is not None else -1


def _bank_lookup(key, w, nfloats, device):
    """-> (_Bank, fresh): the cached bank of this key (created on first use) and whether its contents
    match the weight's current version / epoch."""
    ent = _PACK_CACHE.get(key)
    if ent is not None and ent.w is not w:
        # a fresh VIEW of the same arena parameter (ops.linear reshapes its weight on every call): same address, same
        # shape (both in the key), same epoch cell -> the same bank; anything else at that address is a new tensor
        cell = getattr(w, "_og_epoch", None)
        if cell is not None and cell is getattr(ent.w, "_og_epoch", None):
            ent.w = w
        elif (cell is None and getattr(ent.w, "_og_epoch", None) is None and not w.requires_grad
              and ent.w.untyped_storage().data_ptr() == w.untyped_storage().data_ptr()):
            # an alias of the cached frozen weight (`.detach()`, a re-made view): the entry keeps that storage alive,
            # so the same address is the same memory; in-place edits show in the shared version counter
            ent.w = w
    if ent is None or ent.w is not w:
        if len(_PACK_CACHE) >= _PACK_CACHE_MAX:
            invalidate_packed()
        ent = _Bank(w, torch.zeros(nfloats, dtype=_F32, device=device))   # holding w keeps its address from being reused
        _PACK_CACHE[key] = ent
        cell = getattr(w, "_og_epoch", None)
        if cell is not None:
            grp = _ARENA_BANKS.setdefault(id(cell), {"cell": cell, "banks": [], "table": None, "n": 0})
            grp["banks"].append(ent)
            grp["table"] = None
        return ent, False
    return ent, (ent.version == w._version and ent.epoch == _epoch_of(w))


def _bank_mark(ent):
    ent.version, ent.epoch = ent.w._version, _epoch_of(ent.w)


def repack_arena(epoch_cell):
    """Refresh every cached bank of the network that owns `epoch_cell` (call right after its optimizer
    step, on the same stream): one launch over the device table of pack jobs."""
    grp = _ARENA_BANKS.get(id(epoch_cell))
    if grp is None or not grp["banks"]:
        return 0
    if grp["table"] is None:
        blob = b"".join(j for ent in grp["banks"] for j in ent.jobs)
        nbytes = _q("objgan_conv_pack_job_bytes")
        grp["n"] = len(blob) // nbytes
        dev = grp["banks"][0].wt.device
        grp["table"] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    _lib.call("objgan_conv_pack_jobs_run", _p(grp["table"]), grp["n"], _stream())
    for ent in grp["banks"]:
        _bank_mark(ent)
    return grp["n"]


# Census of the fp16x2 operands (tests; `ops._H2_CENSUS = {}` switches it on, host syncs included).  fp16x2's scale is per
# tensor: a GEMM row / column whose operand entries sit 2^-k below the tensor maximum keeps a relative error of ~2^(k-38)
# (tests/test_kernels_gpu.py::test_fp16x2_per_row_error_follows_the_documented_bound).  The census records, per kind of
# operand, the largest spread log2(max over channels / min over non-zero channels of the per-channel maximum) seen.
_H2_CENSUS = None

# Run-time guard of fp16x2.  The scale of an operand is per TENSOR; a GEMM row / column whose entries sit
# 2^-k below the tensor's maximum keeps ~2^(k - 38) relative error -- harmless while channels stay within 2^16 of each other
# (census of a training step: <= 2^6.2 at B = 16), wrong if training ever drives one channel's scale far from the rest.
# `h2_guard_begin()` .. `h2_guard_end()` bracket ONE checked step (the trainer does that every `h2_guard_every` steps, rank-local,
# host syncs included -- a census step is slower and is not a timed step): every fp16x2 launch reports the per-channel spread
# of the operands whose channels are GEMM rows / columns, keyed by its call SITE (filter-bank address + direction; layer
# geometry for the weight gradient).  A site whose spread exceeds 2^limit is DEMOTED: from then on it runs bf16x3 (exact
# three-way split, no scale), and the event is logged.  Demotions are sticky for the life of the process.
_H2_GUARD = {"on": False, "limit": 16.0, "demoted": {}, "log": [], "checked": 0}


def h2_guard_begin(limit=None):
    global _H2_CENSUS
    if limit is not None:
        _H2_GUARD["limit"] = float(limit)
    _H2_GUARD["on"] = True
    _H2_GUARD["saved_census"] = _H2_CENSUS
    if _H2_CENSUS is None:
        _H2_CENSUS = {}


def h2_guard_end():
    """-> the demotions this checked step added: [(site, kind, spread as log2, operand shape)]"""
    global _H2_CENSUS
    _H2_GUARD["on"] = False
    _H2_CENSUS = _H2_GUARD.pop("saved_census", None)
    new = _H2_GUARD.pop("new", [])
    return new


def h2_guard_state():
    return {"demoted_sites": len(_H2_GUARD["demoted"]), "log": list(_H2_GUARD["log"]), "launches_checked": _H2_GUARD["checked"]}


def h2_guard_reset():
    _H2_GUARD["demoted"].clear()
    del _H2_GUARD["log"][:]
    _H2_GUARD["checked"] = 0


def _census(kind, t, dim, site=None):
    if _H2_CENSUS is None:
        return
    m = t.detach().abs().amax(dim=[d for d in range(t.dim()) if d != dim])
    nz = m[m > 0]
    if nz.numel() == 0:
        return
    spread = float(torch.log2(nz.max() / nz.min()))
    ent = _H2_CENSUS.setdefault(kind, [0.0, None, 0])
    ent[2] += 1
    if spread > ent[0]:
        ent[0], ent[1] = spread, tuple(t.shape)
    if _H2_GUARD["on"] and site is not None:
        _H2_GUARD["checked"] += 1
        if spread > _H2_GUARD["limit"] and site not in _H2_GUARD["demoted"]:
            _H2_GUARD["demoted"][site] = (kind, spread)
            rec = (site, kind, round(spread, 1), tuple(t.shape))
            _H2_GUARD["log"].append(rec)
            _H2_GUARD.setdefault("new", []).append(rec)
            import warnings
            warnings.warn("objgan_hip fp16x2 guard: %s -- per-channel spread 2^%.1f > 2^%.0f on an operand of shape %s; this "
                          "call site runs bf16x3 from now on" % (kind, spread, _H2_GUARD["limit"], tuple(t.shape)))


def _demoted(site):
    return bool(_H2_GUARD["demoted"]) and site in _H2_GUARD["demoted"]


# Debug mode OBJGAN_H2_VERIFY=1: every maximum handed to a kernel from the per-tensor cache / a producer's
# attachment is re-derived by a pass over the tensor and compared on the host -- the cache is keyed on torch's version
# counter, which a kernel writing through a raw pointer does not bump; a stale entry would scale wrongly.  The GPU suite
# runs one training step in this mode (tests/test_modules_gpu.py::test_train_step_hands_no_stale_maxima_to_the_kernels).
_H2_VERIFY = {"on": _os.environ.get("OBJGAN_H2_VERIFY") == "1", "checked": 0}


def _verify_amax(t, amax):
    fresh = torch.empty(_AMAX_SLOTS, dtype=_F32, device=t.device)
    src = t if (t.is_contiguous() and not (t.data_ptr() & 15)) else t.contiguous().clone()
    _lib.call("objgan_absmax_partials", _p(src), src.numel(), _p(fresh), _stream())
    a, b = float(amax.max()), float(fresh.max())
    _H2_VERIFY["checked"] += 1
    if a != b:
        raise _lib.ObjganHipError("stale maximum: cached %.9g, tensor has %.9g (shape %s)" % (a, b, tuple(t.shape)))


_PACK_LOG = {} if _os.environ.get("OBJGAN_PACK_LOG") == "1" else None
if _PACK_LOG is not None:
    import atexit as _atexit

    def _dump_pack_log():
        import sys
        for key, n in sorted(_PACK_LOG.items(), key=lambda kv: -kv[1]):
            sys.stderr.write("PACKLOG %6d x %s\n" % (n, key))
    _atexit.register(_dump_pack_log)


def _pack_log(kind, w, cached):
    """development: which calls still pack their bank themselves (a pack_weights_kernel launch per call)?"""
    import traceback
    where = " < ".join("%s:%d" % (f.name, f.lineno) for f in traceback.extract_stack(limit=9)[:-2][::-1][:6])
    key = "%s %s cached=%s  %s" % (kind, tuple(w.shape), cached, where)
    _PACK_LOG[key] = _PACK_LOG.get(key, 0) + 1


def _job_blob():
    return ctypes.create_string_buffer(_q("objgan_conv_pack_job_bytes"))


# bf16 mode (BASELINE config 5): the matrix kernels read their pixel operand from a bf16 channel-blocked copy [N][C/16][H][W][16] of the source
# that the library writes into the call's workspace.  Tests switch this off to compare with the fp32-gather form of the
# same arithmetic (bit-identical results: same values, same summation order).
_BF16_CHANNELS_LAST = True


def _nhwc_floats(N, C, H, W):
    return (N * H * W * ((C + 15) // 16 * 16) // 2 + 3) & ~3


def _igemm(x, w, bias, y, N, C, H, W, upsample, pad_mode, Cout, Cin, Torig, transpose,
           dh, dw, src_tap, PH, PW, stride, OHf, OWf, osh, osw, ooh, oow, act, y_prezeroed=0, ring=None, cache=True,
           ymax=None):
    Tg = len(dh)
    M = Cin if transpose else Cout
    math = _call_math(2.0 * M * C * Tg * N * PH * PW)
    site = ("conv", w.data_ptr(), int(transpose))
    if math == 4 and _demoted(site):
        math = 2                              # fp16x2 guard: this filter bank's rows / columns are too far apart for one scale
    # the library picks the kernel -- hence the bank layout -- from sizes, taps and math: ask it
    layout = _q("objgan_conv_bank_layout", N, C, H, W, M, Tg, PH, PW, act, math)
    if math == 4 and (layout & 255) != 5:
        math = 2                              # thin / first-generation kernels: no fp16x2 form
        layout = _q("objgan_conv_bank_layout", N, C, H, W, M, Tg, PH, PW, act, math)
    xmax = _absmax(x) if math == 4 else None
    if math == 4 and _H2_CENSUS is not None:
        _census("data gradient: filter columns (input channels)" if transpose else "forward: filter rows (output channels)",
                w, 1 if transpose else 0, site)
    kmath, xk = math, x
    if math == 1 and _B16["on"] and _BF16_CHANNELS_LAST and (layout & 255) == 3:
        xc = _bf16_copy(x, N, C, H * W)
        if xc is not None:
            kmath, xk = 3, xc
    if math == 4 and _REC["on"] and not (x.data_ptr() & 15):
        # same arithmetic, same bank; the pixel operand as its fp16 record where that pays
        rec = _records(x, xmax, N, C, H * W, 2.0 * M * Tg * PH * PW / float(H * W), M)
        if rec is not None:
            kmath, xk = 5, rec
    key = _pack_key(w, transpose, src_tap, layout, math) if cache else None      # (temporaries: pack per call, keep nothing)
    nfl = _q("objgan_conv_packed_floats", int(M), int(C), int(Tg))
    if key is not None:
        ent, fresh = _bank_lookup(key, w, nfl, x.device)
        wt, packed = ent.wt, int(fresh)
        if ent.jobs is None:
            job = _job_blob()
            _lib.call("objgan_conv_pack_job", job, _p(w), _p(wt), N, C, H, W, Cout, Cin, Torig, int(transpose),
                      Tg, _iarr(src_tap), PH, PW, act, math)
            ent.jobs = [job.raw]
        _bank_mark(ent)
    else:
        wt, packed = torch.empty(nfl, dtype=_F32, device=x.device), 0
    if _PACK_LOG is not None and not packed:
        _pack_log("igemm", w, key is not None)
    # split-K launches (small grids, long reductions) go through a workspace: partial tiles, then an ordered sum
    nws = _q("objgan_conv_igemm_ws_floats", N, C, H, W, int(upsample), int(pad_mode), Cout, Cin, Torig,
                                                   int(transpose), Tg, PH, PW, stride, OHf, OWf, osh, osw, act,
                                                   int(y_prezeroed), kmath, 0 if ring is None else 1)
    ws = torch.empty(nws, dtype=_F32, device=x.device) if nws > 0 else None
    if not _BF16_CHANNELS_LAST and math == 1 and nws == _nhwc_floats(N, C, H, W):
        ws, nws = None, 0                 # (tests) no workspace: the library gathers from the fp32 NCHW source instead
    _lib.call("objgan_conv_igemm", _p(xk), _p(w), _p(bias), _p(y), _p(wt),
              N, C, H, W, int(upsample), int(pad_mode), Cout, Cin, Torig, int(transpose),
              Tg, _iarr(dh), _iarr(dw), _iarr(src_tap), PH, PW, stride,
              OHf, OWf, osh, osw, ooh, oow, act, int(y_prezeroed), packed, kmath, _p(ring), _p(xmax), _p(ymax),
              _p(ws), nws, _stream())


def _dgrad_s2_phases(g, w, N, Cout, OH, OW, Cin, k, pad_h, pad_w, LH, LW, cacheable=True):
    """dX of a stride-2 conv in ONE launch when every output parity phase has the same tap count
    (k even, even sizes); returns None when the shape does not qualify."""
    KH, KW = (k, k) if isinstance(k, int) else k
    if KH % 2 or KW % 2 or LH % 2 or LW % 2 or Cin <= 32:
        return None
    if float(N) * Cout * OH * OW * 4.0 >= 4.0e9:
        return None
    dh, dw, st = [], [], []
    for a in range(2):
        khs = [kh for kh in range(KH) if (a + pad_h - kh) % 2 == 0]
        for b in range(2):
            kws = [kw for kw in range(KW) if (b + pad_w - kw) % 2 == 0]
            dh += [(a + pad_h - kh) // 2 for kh in khs for kw in kws]
            dw += [(b + pad_w - kw) // 2 for kh in khs for kw in kws]
            st += [kh * KW + kw for kh in khs for kw in kws]
    Tg = len(st) // 4
    if Tg > 8:
        return None
    dx = torch.empty((N, Cin, LH, LW), dtype=_F32, device=g.device)
    n = 4 * ((Cin * Tg * ((Cout + 15) // 16 * 16) * 3 + 1) // 2) + _AMAX_SLOTS    # bf16x3 banks: 6 B per element; + |w| maxima
    math = _call_math(2.0 * Cin * Cout * Tg * N * LH * LW)
    site = ("conv", w.data_ptr(), 1)
    if math == 4 and _demoted(site):
        math = 2
    xmax = _absmax(g) if math == 4 else None
    if math == 4 and _H2_CENSUS is not None:
        _census("data gradient: filter columns (input channels)", w, 1, site)
    key = _pack_key(w, 2, st, False, math) if cacheable else None
    if key is not None:
        ent, fresh = _bank_lookup(key, w, n, g.device)
        wt, packed = ent.wt, int(fresh)
        if ent.jobs is None:
            ent.jobs = []
            for ph in range(4):
                job = _job_blob()
                _lib.call("objgan_conv_pack_job_phase", job, _p(w), _p(wt), Cout, Cin, KH * KW, Tg,
                          _iarr(st[ph * Tg:(ph + 1) * Tg]), ph, math)
                ent.jobs.append(job.raw)
        _bank_mark(ent)
    else:
        wt, packed = torch.empty(n, dtype=_F32, device=g.device), 0
    if _PACK_LOG is not None and not packed:
        _pack_log("phases", w, key is not None)
    nws = _q("objgan_conv_dgrad_s2_phases_ws_floats", N, Cout, OH, OW, math)
    ws = torch.empty(nws, dtype=_F32, device=g.device) if nws > 0 and _BF16_CHANNELS_LAST else None
    nws = nws if ws is not None else 0
    kmath, gk = math, g
    if math == 4 and _REC["on"] and not (g.data_ptr() & 15):
        rec = _records(g, xmax, N, Cout, OH * OW, 2.0 * Cin * Tg * LH * LW / float(OH * OW), Cin)
        if rec is not None:
            kmath, gk = 5, rec
    _lib.call("objgan_conv_dgrad_s2_phases", _p(gk), _p(w), _p(dx), _p(wt), N, Cout, OH, OW, Cin, KH * KW,
              Tg, _iarr(dh), _iarr(dw), _iarr(st), LH // 2, LW // 2, packed, kmath, _p(xmax), _p(ws), nws, _stream())
    return dx


_THIN4 = {"on": _os.environ.get("OBJGAN_THIN4", "1") != "0"}


def _dgrad_s2_thin(g, w, N, Cout, OH, OW, Cin, cacheable=True):
    """dX of a 4x4 / stride-2 / pad-1 convolution w.r.t. <= 32 input channels (layout code / image part of the first
    shape / object discriminator convolution): the four output parity phases in ONE launch of the fp32 VALU kernel -- dY is
    read twice instead of once per phase."""
    dx = torch.empty((N, Cin, 2 * OH, 2 * OW), dtype=_F32, device=g.device)
    n = _q("objgan_conv_dgrad_s2_thin_floats", Cout, Cin)
    key = _pack_key(w, 3, (0,), "thin4", 0) if cacheable else None
    if key is not None:
        ent, fresh = _bank_lookup(key, w, n, g.device)
        wt, packed = ent.wt, int(fresh)
        if ent.jobs is None:
            ent.jobs = []
            for ph in range(4):
                job = _job_blob()
                _lib.call("objgan_conv_pack_job_thin_phase", job, _p(w), _p(wt), Cout, Cin, ph)
                ent.jobs.append(job.raw)
        _bank_mark(ent)
    else:
        wt, packed = torch.empty(n, dtype=_F32, device=g.device), 0
    if _PACK_LOG is not None and not packed:
        _pack_log("thin4", w, key is not None)
    _lib.call("objgan_conv_dgrad_s2_thin", _p(g), _p(w), _p(dx), _p(wt), N, Cout, OH, OW, Cin, packed, _stream())
    return dx


def conv_out_size(L, k, s, p):
    return (L + 2 * p - k) // s + 1


def _conv_fwd(x, w, bias, stride, pad, refl, upsample, act):
    """y = act(conv(x, w) + bias); x [N, Cin, H, W] (nearest x2 lifted when `upsample`), w [Cout, Cin, k, k]."""
    N, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    LH, LW = (2 * H, 2 * W) if upsample else (H, W)
    OH, OW = conv_out_size(LH, k, stride, pad), conv_out_size(LW, k, stride, pad)
    y = torch.empty((N, Cout, OH, OW), dtype=_F32, device=x.device)
    dh = [kh - pad for kh in range(k) for kw in range(k)]
    dw = [kw - pad for kh in range(k) for kw in range(k)]
    st = list(range(k * k))
    # a convolution with a fused LeakyReLU / ReLU feeds the next convolution directly (discriminator encoders): its
    # epilogue leaves the partial maxima of y for that consumer's fp16x2 scale
    ym = _amax_zeroed(x.device) if (act in ("lrelu", "relu") and _amax_wanted(y.numel())) else None
    _igemm(x, w, bias, y, N, Cin, H, W, upsample, refl, Cout, Cin, k * k, 0,
           dh, dw, st, OH, OW, stride, OH, OW, 1, 1, 0, 0, _ACT[act], ymax=ym)
    if ym is not None:
        _amax_attach(y, ym)
    return y


def _conv_dgrad(g, w, N, Cin, H, W, stride, pad, refl, upsample, cacheable=True):
    """Gradient w.r.t. the input [N, Cin, H, W] of conv(x, w) given g = dL/d(conv output)."""
    Cout, k = w.shape[0], w.shape[2]
    OH, OW = g.shape[2], g.shape[3]
    LH, LW = (2 * H, 2 * W) if upsample else (H, W)
    if stride == 1:
        pe = 0 if refl else pad            # reflect: gradient w.r.t. the padded tensor first
        TH, TW = (LH + 2 * pad, LW + 2 * pad) if refl else (LH, LW)
        dh = [pe - kh for kh in range(k) for kw in range(k)]
        dw = [pe - kw for kh in range(k) for kw in range(k)]
        st = list(range(k * k))
        ring_ok = (refl and pad == 1 and LH >= 3 and LW >= 3 and
                   (_q("objgan_conv_bank_layout", N, Cout, OH, OW, Cin, k * k, TH, TW, 0, _MATH["mode"]) & 255) in (1, 3, 4, 5))
        if ring_ok:
            # gradient of the reflect-padded conv without the padded intermediate: interior pixels go straight
            # into dX, the one-pixel border into a small ring buffer that is mirrored back afterwards
            dxl = torch.empty((N, Cin, LH, LW), dtype=_F32, device=g.device)
            ring = torch.empty((N * Cin, 2 * TW + 2 * TH), dtype=_F32, device=g.device)
            _igemm(g, w, None, dxl, N, Cout, OH, OW, 0, 0, Cout, Cin, k * k, 1,
                   dh, dw, st, TH, TW, 1, LH, LW, 1, 1, 0, 0, 0, ring=ring, cache=cacheable)
            _lib.call("objgan_reflect_ring_fold", _p(ring), _p(dxl), N * Cin, LH, LW, _stream())
        else:
            dxl = torch.empty((N, Cin, TH, TW), dtype=_F32, device=g.device)
            _igemm(g, w, None, dxl, N, Cout, OH, OW, 0, 0, Cout, Cin, k * k, 1,
                   dh, dw, st, TH, TW, 1, TH, TW, 1, 1, 0, 0, 0, cache=cacheable)
            if refl:
                folded = torch.empty((N, Cin, LH, LW), dtype=_F32, device=g.device)
                _lib.call("objgan_reflect_fold", _p(dxl), _p(folded), N * Cin, LH, LW, _stream())
                dxl = folded
    elif stride == 2:
        if refl:
            raise _lib.ObjganHipError("stride-2 reflect conv is not on the hot path")
        dxl = None
        if (_THIN4["on"] and k == 4 and pad == 1 and Cin <= 12 and LH == 2 * OH and LW == 2 * OW
                and N * OH * OW >= 65536 and float(N) * Cout * OH * OW * 4.0 < 4.0e9):
            dxl = _dgrad_s2_thin(g, w, N, Cout, OH, OW, Cin, cacheable)      # (thin outputs: the four phases in one launch)
        if dxl is None:
            dxl = _dgrad_s2_phases(g, w, N, Cout, OH, OW, Cin, k, pad, pad, LH, LW, cacheable)
        phases = range(2) if dxl is None else ()
        if dxl is None:
            dxl = torch.zeros((N, Cin, LH, LW), dtype=_F32, device=g.device)
        for ph in phases:
            khs = [kh for kh in range(k) if (ph + pad - kh) % 2 == 0]
            PHg = (LH - ph + 1) // 2
            for pw in range(2):
                kws = [kw for kw in range(k) if (pw + pad - kw) % 2 == 0]
                PWg = (LW - pw + 1) // 2
                if PHg <= 0 or PWg <= 0:
                    continue
                dh = [(ph + pad - kh) // 2 for kh in khs for kw in kws]
                dw = [(pw + pad - kw) // 2 for kh in khs for kw in kws]
                st = [kh * k + kw for kh in khs for kw in kws]
                if not st:      # no tap reaches this phase: gradient is zero there
                    dh, dw, st = [0], [0], [-1]
                _igemm(g, w, None, dxl, N, Cout, OH, OW, 0, 0, Cout, Cin, k * k, 1,
                       dh, dw, st, PHg, PWg, 1, LH, LW, 2, 2, ph, pw, 0, y_prezeroed=1, cache=cacheable)
    else:
        raise _lib.ObjganHipError("conv2d backward: stride %d not supported" % stride)
    if upsample:
        dx = torch.empty((N, Cin, H, W), dtype=_F32, device=g.device)
        _lib.call("objgan_sum2x2", _p(dxl), _p(dx), N * Cin, H, W, _stream())
        return dx
    return dxl


# Weight gradients beside the data-gradient chain (round 6).  A backward pass is a serial chain of data gradients -- each
# layer's needs the next layer's -- interleaved with normalisation / activation derivatives that do not fill 256 CUs; the
# weight gradient of a layer feeds nothing but the optimizer.  Inside `wgrad_stream_scope(side)` (the trainer: around the
# generator's backward pass, the one phase of the step that runs on a single stream) a weight gradient that goes straight
# into its arena (`sink`) is LAUNCHED ON `side` behind an event of the issuing stream, and the issuing stream goes on with
# the data gradient; whoever reads the gradient arena next (ArenaAdam.step, an all-reduce) calls `wgrad_join()` first.
# Scale inputs / records are produced (or found in the per-stream caches) on the issuing stream BEFORE the hand-over, the
# split workspace is allocated under `side`, and every tensor the launch reads is recorded on `side` for the caching
# allocator.  Same kernels, same arguments: bit-identical gradients.
_WG_ASYNC = {"stream": None, "pending": []}


class wgrad_stream_scope(object):
    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        self.prev = _WG_ASYNC["stream"]
        _WG_ASYNC["stream"] = self.stream
        return self

    def __exit__(self, *exc):
        _WG_ASYNC["stream"] = self.prev
        return False


def wgrad_join():
    """the current stream waits for every weight gradient that was handed to a side stream"""
    if _WG_ASYNC["pending"]:
        cur = torch.cuda.current_stream()
        for st in _WG_ASYNC["pending"]:
            cur.wait_stream(st)
        del _WG_ASYNC["pending"][:]


def _conv_wgrad(x, g, Cout, k, stride, pad, refl, upsample, sink=None):
    """Gradient w.r.t. the filter bank [Cout, Cin, k, k] of conv(x, w) given g = dL/d(conv output).
    sink: the weight's (gradient view, notify) pair from its optimizer arena (`_grad_sink`): the ordered combine of the
    splits ADDS its sum into the view (the same `grad + dw` autograd's AccumulateGrad would compute, bit for bit, without
    the temporary and the add launch) and None is returned -- the caller hands None to autograd and calls notify()."""
    N, Cin, H, W = x.shape
    if sink is not None:
        dw_ = sink[0]
    else:
        dw_ = torch.empty((Cout, Cin, k, k), dtype=_F32, device=x.device)      # fully written: no zero-fill
    math = _call_math(2.0 * Cout * Cin * k * k * N * g.shape[2] * g.shape[3])
    site = ("wgrad", sink[0].data_ptr() if sink is not None else 0, N, Cin, H, W, Cout, k, stride, int(upsample), refl)
    if math == 4 and _demoted(site):
        math = 2                              # fp16x2 guard: x / dy channels too far apart for one scale per tensor
    geo = (N, Cin, H, W, int(upsample), refl, Cout, g.shape[2], g.shape[3], k, stride, pad, math)
    xmax, gmax = (_absmax(x), _absmax(g)) if math == 4 else (None, None)
    if math == 4 and _H2_CENSUS is not None:
        _census("weight gradient: x channels (filter columns)", x, 1, site)
        _census("weight gradient: dy channels (filter rows)", g, 1, site)
    xk = x
    # Where: measured per shape class (profiles/r05_records_convbench.txt).  The record form wins where the register-
    # fragment gather has no constant-stride fast path or runs on few pixels -- up-sampled sources 120 -> 257 TFLOP/s,
    # reflect-padded maps up to 64 x 64: 85 -> 137 at 32 x 32 -- ties on the large stride-1 / stride-2 layers (254 vs 258,
    # 251 vs 269) and loses where it has to give up the 7-group block row (194 -> 194 at 128 x 128: 239 vs 214) or runs
    # 4-wave workgroups at one wave per SIMD (16 x 16 maps, 186 vs 129): those keep the gather form.
    rec_pays = bool(upsample) or (refl and H * W <= 4096) or _REC["wgrad"] == "all"
    if (math == 4 and _REC["on"] and _REC["wgrad"] and rec_pays and not (x.data_ptr() & 15) and
            _q("objgan_conv_wgrad_rec_ok", N, Cin, H, W, Cout, g.shape[2], g.shape[3], k)):
        # x as its fp16 record (usually the one the forward convolution of this layer made): half-record loads +
        # transposing LDS reads instead of 32-plane gathers and the split on the VALU
        rec = _records(x, xmax, N, Cin, H * W, 2.0 * Cout * k * k * g.shape[2] * g.shape[3] / float(H * W), Cout)
        if rec is not None:
            geo = geo[:-1] + (_REC["wgrad_math"],)      # 6: dy pre-split too (its fp16 pair, one pass into the workspace); 5: split in the loop
            xk = rec
    if (math == 1 and _B16["on"] and _BF16_CHANNELS_LAST
            and _q("objgan_conv_wgrad_bfb_ok", N, Cin, H, W, Cout, g.shape[2], g.shape[3], k)):
        xc = _bf16_copy(x, N, Cin, H * W)            # (usually the copy the forward convolution of this layer made)
        if xc is not None:
            geo = geo[:-1] + (3,)
            xk = xc
    nws = _q("objgan_conv_wgrad_ws_floats", *geo)
    side = _WG_ASYNC["stream"] if (sink is not None and _H2_CENSUS is None) else None
    if side is not None:
        # hand the launch to the side stream: everything it reads exists on the issuing stream by now
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        for t in (xk, g, xmax, gmax):
            if t is not None:
                t.record_stream(side)
        with torch.cuda.stream(side):
            ws = torch.empty(nws, dtype=_F32, device=x.device) if nws > 0 else None
            _lib.call("objgan_conv_wgrad", _p(xk), _p(g), _p(dw_), *geo, 1, _p(xmax), _p(gmax), _p(ws), nws, _stream())
        if side not in _WG_ASYNC["pending"]:
            _WG_ASYNC["pending"].append(side)
        sink[1]()
        return None
    ws = torch.empty(nws, dtype=_F32, device=x.device) if nws > 0 else None
    _lib.call("objgan_conv_wgrad", _p(xk), _p(g), _p(dw_), *geo, 0 if sink is None else 1, _p(xmax), _p(gmax), _p(ws), nws,
              _stream())
    if sink is not None:
        sink[1]()
        return None
    return dw_


# Weight gradients straight into the optimizer arena.  A parameter that lives in a flat arena (trainer.ParamArena)
# carries `_og_grad_sink = (view of its slot in the gradient arena, notify)`; its `.grad` IS that view.  The convolution
# backward then accumulates into the view itself and returns None to autograd (no temporary, no AccumulateGrad `add`
# launch: 537 of them per training step in round 3) and calls notify() in AccumulateGrad's place (the arena's
# "gradient is final" bookkeeping for the bucketed all-reduce).  Only under .backward() into .grad -- the arena's
# owner switches the sinks off (`direct_wgrad(False)`) around anything else.
_DIRECT_WGRAD = {"on": False}       # off by default: the arena's owner enables it around its backward passes


def direct_wgrad(on):
    _DIRECT_WGRAD["on"] = bool(on)


class direct_wgrad_scope(object):
    """with ops.direct_wgrad_scope(): weight gradients of arena parameters accumulate straight into their arena views"""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _DIRECT_WGRAD["on"]
        _DIRECT_WGRAD["on"] = self.on
        return self

    def __exit__(self, *exc):
        _DIRECT_WGRAD["on"] = self.prev
        return False


def _grad_sink(w, shape):
    sink = getattr(w, "_og_grad_sink", None) if _DIRECT_WGRAD["on"] else None
    if sink is None or w.grad is None or w.grad.data_ptr() != sink[0].data_ptr() or tuple(sink[0].shape) != tuple(shape):
        return None
    return sink


def _channel_sum(g, out, N, C, HW):
    """out[c] = sum over images and pixels of g[:, c] (bias gradient), partial sums combined in order"""
    n = _lib.load().objgan_channel_sum_ws_floats(N, C, HW)
    ws = torch.empty(n, dtype=_F32, device=g.device) if n > 0 else None
    _lib.call("objgan_channel_sum", _p(g), _p(out), N, C, HW, _p(ws), _stream())


class _Conv2dFn(torch.autograd.Function):
    """conv2d with the gather-side fusions of the hot path.

    pad_mode 'zeros' : nn.Conv2d(padding=pad)            (reference model.py:36-39, 1002, 1010)
    pad_mode 'reflect': nn.ReflectionPad2d(pad) + nn.Conv2d(padding=0)  (model.py:66-75, 599-601)
    upsample          : nn.Upsample(scale_factor=2, mode='nearest') in front (model.py:43-49)
    act               : LeakyReLU(0.2) / tanh / sigmoid fused in the epilogue
    """

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, pad_mode, upsample, act):
        _chk(x, w, bias)
        x = _c(x)
        w = _c(w)
        if w.shape[1] != x.shape[1] or w.shape[2] != w.shape[3]:
            raise _lib.ObjganHipError("conv2d: bad weight shape %s for input %s" % (tuple(w.shape), tuple(x.shape)))
        refl = 1 if pad_mode == "reflect" else 0
        if refl and pad != 1:
            raise _lib.ObjganHipError("reflect padding is implemented for pad=1")
        y = _conv_fwd(x, w, bias, stride, pad, refl, upsample, act)
        ctx.cfg = (stride, pad, refl, bool(upsample), act, w.shape[2])
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w, y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, refl, upsample, act, k = ctx.cfg
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dy = _c(dy)
        _chk(dy)
        if act not in (None, "none"):
            g = torch.empty_like(dy)
            am = torch.empty(_AMAX_SLOTS, dtype=_F32, device=dy.device) if _amax_wanted(dy.numel()) else None
            _lib.call("objgan_act_backward", _p(dy), _p(y), _p(g), dy.numel(), _ACT[act], _p(am), _stream())
            if am is not None:
                _amax_attach(g, am)
        else:
            g = dy
        dx = dw_ = db = None
        if ctx.needs_input_grad[1]:             # (first: inside wgrad_stream_scope it leaves for a side stream at once)
            dw_ = _conv_wgrad(x, g, Cout, k, stride, pad, refl, upsample, sink=_grad_sink(w, (Cout, Cin, k, k)))
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad(g, w, N, Cin, H, W, stride, pad, refl, upsample)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(Cout, dtype=_F32, device=x.device)
            _channel_sum(g, db, N, Cout, g.shape[2] * g.shape[3])
        return dx, dw_, db, None, None, None, None, None


# ---- nearest x2 upsample + 3x3 conv (upBlock, reference model.py:43-49) as its four output-parity
# phases.  On the lifted map the three taps of an axis read only two distinct source rows:
#   y[2i]   = w0 x[i-1] + (w1 + w2) x[i]        y[2i+1] = (w0 + w1) x[i] + w2 x[i+1]
# so every phase is a 2x2 convolution of the LOW-resolution input with pre-summed filters: 16
# MACs per source pixel and channel pair instead of 36.  Written with a 4x4 bank W4 = A w A^T
# (A below), y is exactly the TRANSPOSED stride-2 4x4 convolution of x -- the one-launch phased
# data-gradient kernel -- and the gradients are the forward / weight gradient of that stride-2
# convolution: three kernels the discriminators already use, nothing new on the device.  The only
# arithmetic difference to the lifted form is the pre-summation of filter taps (re-association at
# the 1e-7 level).  fp32 / bf16x3 math only: the bf16 mode is defined on the rounded ORIGINAL filters.
_UP_A = ((0., 0., 1.), (0., 1., 1.), (1., 1., 0.), (1., 0., 0.))
_UP4 = {}           # (weight address, shape) -> [w, _version, epoch, W4]


_UP_MAT = {}


def _up_matrix(device):
    """the constant 4x3 tap-summation matrix, uploaded ONCE per device (a `torch.tensor(..., device=)` per call was a
    synchronous host->device copy: five full host-device synchronisations per training step)"""
    m = _UP_MAT.get(device)
    if m is None:
        m = _UP_MAT[device] = torch.tensor(_UP_A, dtype=_F32, device=device)
    return m


def _up_bank(w):
    """W4 [Cin, Cout, 4, 4] of a [Cout, Cin, 3, 3] filter bank; rebuilt (in place) only when the
    weights changed, so that its packed images stay cached as well."""
    ep = getattr(w, "_og_epoch", None)
    cacheable = ep is not None or not w.requires_grad
    A = _up_matrix(w.device)
    if not cacheable:
        return torch.einsum("pk,mckl,ql->cmpq", A, w.detach(), A).contiguous(), False
    key = (w.data_ptr(), tuple(w.shape))
    ent = _UP4.get(key)
    epv = ep[0] if ep is not None else -1
    if ent is not None and ent[0] is w and ent[1] == w._version and ent[2] == epv:
        return ent[3], True
    W4 = ent[3] if ent is not None and ent[0] is w else torch.empty(
        (w.shape[1], w.shape[0], 4, 4), dtype=_F32, device=w.device)
    W4.copy_(torch.einsum("pk,mckl,ql->cmpq", A, w.detach(), A))      # in place: bumps W4._version
    if len(_UP4) >= 256:
        _UP4.clear()
    _UP4[key] = [w, w._version, epv, W4]
    return W4, True


def _up_phased_ok(x, w, bias, stride, pad, pad_mode, upsample, act):
    return (upsample and stride == 1 and pad == 1 and pad_mode != "reflect" and bias is None
            and act in (None, "none") and w.shape[2] == 3 and w.shape[3] == 3 and w.shape[0] > 32
            and x.shape[1] > 32 and x.shape[2] >= 2 and x.shape[3] >= 2 and _MATH["mode"] != 1
            and float(x.shape[0]) * max(x.shape[1], 4 * w.shape[0]) * x.shape[2] * x.shape[3] * 4.0 < 4.0e9)


class _UpConv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        _chk(x, w)
        x = _c(x)
        w = _c(w)
        N, C, H, W = x.shape
        M = w.shape[0]
        if w.shape[1] != C:
            raise _lib.ObjganHipError("conv2d: bad weight shape %s for input %s" % (tuple(w.shape), tuple(x.shape)))
        W4, cacheable = _up_bank(w)
        # transposed stride-2 conv of x: "input gradient" of the virtual conv [N, M, 2H, 2W] -> [N, C, H, W]
        y = _conv_dgrad(x, W4, N, M, 2 * H, 2 * W, 2, 1, 0, False, cacheable)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        _chk(dy)
        C, M = x.shape[1], w.shape[0]
        dx = dw_ = None
        if ctx.needs_input_grad[0]:
            W4, _ = _up_bank(w)
            dx = _conv_fwd(dy, W4, None, 2, 1, 0, False, None)
        if ctx.needs_input_grad[1]:
            dW4 = _conv_wgrad(dy, x, C, 4, 2, 1, 0, False)             # [C, M, 4, 4]
            A = _up_matrix(x.device)
            dw_ = torch.einsum("pk,cmpq,ql->mckl", A, dW4, A).contiguous()
        return dx, dw_


def _w_slice(w, lo, hi):
    """contiguous copy of w[:, lo:hi] kept while the weights are unchanged (rebuilt IN PLACE when they changed, so that
    its packed banks stay cached as well -- `_up_bank` does the same for the pre-summed up-convolution bank): the
    per-input data gradients of conv2d_cat packed their slice's bank on every call (80 pack launches per training step).
    The cache lives ON the weight object (a process-wide table keyed on the address pinned dead weights and their
    slices until it overflowed): it dies with the weight.
    -> (slice, cacheable)"""
    ep = getattr(w, "_og_epoch", None)
    if ep is None and w.requires_grad:
        return w[:, lo:hi].contiguous(), False
    try:
        cache = w.__dict__.setdefault("_og_wslices", {})
    except AttributeError:
        return w[:, lo:hi].contiguous(), False
    ent = cache.get((lo, hi))
    epv = ep[0] if ep is not None else -1
    if ent is not None and ent[0] == w._version and ent[1] == epv and ent[3] == w.data_ptr():
        return ent[2], True
    ws = ent[2] if (ent is not None and ent[3] == w.data_ptr()) else torch.empty(
        (w.shape[0], hi - lo) + tuple(w.shape[2:]), dtype=_F32, device=w.device)
    ws.copy_(w.detach()[:, lo:hi])                      # in place: bumps ws._version
    cache[(lo, hi)] = [w._version, epv, ws, w.data_ptr()]
    return ws, True


class _Conv2dCatFn(torch.autograd.Function):
    """conv2d(cat([x1, x2], 1), w) with per-input data gradients: the first convolution of the shape / object
    discriminators sees [image (3) | encoded layout (12)] (reference model.py:1121-1128, 1217-1220).  A discriminator
    step needs only the layout part's gradient, a generator step only the image part's -- each is the data gradient
    of a slice of the filter bank (12 resp. 3 of the 15 input channels), not of the whole bank."""

    @staticmethod
    def forward(ctx, x1, x2, w, stride, pad, act):
        _chk(x1, x2, w)
        x = torch.cat([x1, x2], dim=1)
        w = _c(w)
        if w.shape[1] != x.shape[1] or w.shape[2] != w.shape[3]:
            raise _lib.ObjganHipError("conv2d_cat: bad weight shape %s for inputs %s + %s"
                                      % (tuple(w.shape), tuple(x1.shape), tuple(x2.shape)))
        y = _conv_fwd(x, w, None, stride, pad, 0, False, act)
        ctx.cfg = (stride, pad, act, w.shape[2], x1.shape[1])
        ctx.save_for_backward(x, w, y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, act, k, c1 = ctx.cfg
        N, C, H, W = x.shape
        dy = _c(dy)
        _chk(dy)
        if act not in (None, "none"):
            g = torch.empty_like(dy)
            am = torch.empty(_AMAX_SLOTS, dtype=_F32, device=dy.device) if _amax_wanted(dy.numel()) else None
            _lib.call("objgan_act_backward", _p(dy), _p(y), _p(g), dy.numel(), _ACT[act], _p(am), _stream())
            if am is not None:
                _amax_attach(g, am)
        else:
            g = dy
        n1, n2, nw = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx1 = dx2 = dw_ = None
        if n1 and n2:
            dx = _conv_dgrad(g, w, N, C, H, W, stride, pad, 0, False)
            dx1, dx2 = dx[:, :c1], dx[:, c1:]
        elif n1:
            ws, ok = _w_slice(w, 0, c1)
            dx1 = _conv_dgrad(g, ws, N, c1, H, W, stride, pad, 0, False, cacheable=ok)
        elif n2:
            ws, ok = _w_slice(w, c1, C)
            dx2 = _conv_dgrad(g, ws, N, C - c1, H, W, stride, pad, 0, False, cacheable=ok)
        if nw:
            dw_ = _conv_wgrad(x, g, w.shape[0], k, stride, pad, 0, False, sink=_grad_sink(w, tuple(w.shape)))
        return dx1, dx2, dw_, None, None, None


def conv2d_cat(x1, x2, w, stride=1, pad=0, act=None):
    """act(conv2d(cat([x1, x2], 1), w)) (zero padding, no bias) with per-input data gradients."""
    return _Conv2dCatFn.apply(x1, x2, w, stride, pad, act)


def conv2d(x, w, bias=None, stride=1, pad=0, pad_mode="zeros", upsample=False, act=None):
    if _up_phased_ok(x, w, bias, stride, pad, pad_mode, upsample, act):
        return _UpConv3x3Fn.apply(x, w)
    return _Conv2dFn.apply(x, w, bias, stride, pad, pad_mode, upsample, act)


class _ConvFrozenFn(torch.autograd.Function):
    """Convolution with a FROZEN (non-trainable) rectangular filter bank: forward + input gradient
    only.  Serves the frozen Inception-v3 encoder (1x1, 3x3, 5x5, 1x7, 7x1, 1x3, 3x1 kernels,
    stride 1 or 2, asymmetric zero padding) on the same MFMA implicit-GEMM kernel."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, ph, pw, act):
        _chk(x, w, bias)
        if w.requires_grad or (bias is not None and bias.requires_grad):
            raise _lib.ObjganHipError("conv2d_frozen: the filter bank must not require gradients")
        x, w = _c(x), _c(w)
        N, Cin, H, W = x.shape
        Cout, Cin2, KH, KW = w.shape
        if Cin2 != Cin or KH * KW > 32:
            raise _lib.ObjganHipError("conv2d_frozen: unsupported filter %s" % (tuple(w.shape),))
        OH = (H + 2 * ph - KH) // stride + 1
        OW = (W + 2 * pw - KW) // stride + 1
        y = torch.empty((N, Cout, OH, OW), dtype=_F32, device=x.device)
        dh = [kh - ph for kh in range(KH) for kw in range(KW)]
        dw = [kw - pw for kh in range(KH) for kw in range(KW)]
        # (the frozen encoder's convolutions feed each other through a fused ReLU: the epilogue leaves the maxima)
        ym = _amax_zeroed(x.device) if (act in ("lrelu", "relu") and _amax_wanted(y.numel())) else None
        _igemm(x, w, bias, y, N, Cin, H, W, 0, 0, Cout, Cin, KH * KW, 0, dh, dw, list(range(KH * KW)),
               OH, OW, stride, OH, OW, 1, 1, 0, 0, _ACT[act], ymax=ym)
        if ym is not None:
            _amax_attach(y, ym)
        ctx.cfg = (stride, ph, pw, act, KH, KW, (N, Cin, H, W))
        ctx.save_for_backward(w, y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        w, y = ctx.saved_tensors
        stride, ph, pw, act, KH, KW, (N, Cin, H, W) = ctx.cfg
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None, None
        Cout = w.shape[0]
        dy = _c(dy)
        OH, OW = dy.shape[2], dy.shape[3]
        if act not in (None, "none"):
            g = torch.empty_like(dy)
            am = torch.empty(_AMAX_SLOTS, dtype=_F32, device=dy.device) if _amax_wanted(dy.numel()) else None
            _lib.call("objgan_act_backward", _p(dy), _p(y), _p(g), dy.numel(), _ACT[act], _p(am), _stream())
            if am is not None:
                _amax_attach(g, am)
        else:
            g = dy
        T = KH * KW
        if stride == 1:
            dx = torch.empty((N, Cin, H, W), dtype=_F32, device=dy.device)
            dh = [ph - kh for kh in range(KH) for kw in range(KW)]
            dw = [pw - kw for kh in range(KH) for kw in range(KW)]
            _igemm(g, w, None, dx, N, Cout, OH, OW, 0, 0, Cout, Cin, T, 1, dh, dw, list(range(T)),
                   H, W, 1, H, W, 1, 1, 0, 0, 0)
        elif stride == 2:
            dx = _dgrad_s2_phases(g, w, N, Cout, OH, OW, Cin, (KH, KW), ph, pw, H, W)
            phases = range(2) if dx is None else ()
            if dx is None:
                dx = torch.zeros((N, Cin, H, W), dtype=_F32, device=dy.device)
            for a in phases:
                khs = [kh for kh in range(KH) if (a + ph - kh) % 2 == 0]
                PHg = (H - a + 1) // 2
                for b in range(2):
                    kws = [kw for kw in range(KW) if (b + pw - kw) % 2 == 0]
                    PWg = (W - b + 1) // 2
                    if PHg <= 0 or PWg <= 0 or not khs or not kws:
                        continue
                    dh = [(a + ph - kh) // 2 for kh in khs for kw in kws]
                    dw = [(b + pw - kw) // 2 for kh in khs for kw in kws]
                    st = [kh * KW + kw for kh in khs for kw in kws]
                    _igemm(g, w, None, dx, N, Cout, OH, OW, 0, 0, Cout, Cin, T, 1, dh, dw, st,
                           PHg, PWg, 1, H, W, 2, 2, a, b, 0, y_prezeroed=1)
        else:
            raise _lib.ObjganHipError("conv2d_frozen backward: stride %d not supported" % stride)
        return dx, None, None, None, None, None, None


def conv2d_frozen(x, w, bias=None, stride=1, pad=(0, 0), act=None):
    return _ConvFrozenFn.apply(x, w, bias, int(stride), int(pad[0]), int(pad[1]), act)


def linear(x, w, bias=None, act=None):
    """nn.Linear as a 1x1 convolution over a [N, C, 1, 1] view (reference model.py:462, 494)."""
    w4 = w.reshape(w.shape[0], w.shape[1], 1, 1)
    cell = getattr(w, "_og_epoch", None)
    if cell is not None and w4 is not w:
        w4._og_epoch = cell                  # the view stands for the arena parameter: its packed bank stays cached
    y = conv2d(x.reshape(x.shape[0], x.shape[1], 1, 1), w4, bias, 1, 0, "zeros", False, act)
    return y.reshape(x.shape[0], w.shape[0])


# ----------------------------------------------------------------------------------------------
# normalisation (+ GLU / LeakyReLU / residual)
# ----------------------------------------------------------------------------------------------
_NORM_MODE = {None: 0, "none": 0, "lrelu": 1, "glu": 2}


_LAST_AMAX = [None]         # partial maxima the last _NormActFn.forward left for its output (host-side hand-over)


class _NormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, per_channel, mode, eps,
                momentum):
        _chk(x, gamma, beta, residual, running_mean, running_var)
        x = _c(x)
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C) if x.numel() else 0
        Co = C // 2 if mode == "glu" else C
        y = torch.empty((N, Co) + tuple(x.shape[2:]), dtype=_F32, device=x.device)
        G = C if per_channel else N * C
        nst = _q("objgan_norm_ws_floats", N, C, HW, int(per_channel))     # statistics workspace (ordered combine)
        ws = torch.empty(nst + 2 * G, dtype=_F32, device=x.device)
        sums, mean, rstd = ws[:nst], ws[nst:nst + G], ws[nst + G:]
        residual = _c(residual) if residual is not None else None
        emit = _q("objgan_norm_amax_supported", N, C, HW, int(per_channel), int(gamma is not None)) \
            if _amax_wanted(y.numel()) else 0
        # (2: the one-kernel InstanceNorm adds its maxima atomically into slots the caller zeroed)
        am = (_amax_zeroed(x.device) if emit == 2 else torch.empty(_AMAX_SLOTS, dtype=_F32, device=x.device)) if emit else None
        _lib.call("objgan_norm_forward", _p(x), _p(y), _p(residual), _p(gamma), _p(beta),
                  _p(running_mean), _p(running_var), _p(sums), _p(mean), _p(rstd),
                  N, C, HW, int(per_channel), _NORM_MODE[mode], float(eps), float(momentum), _p(am), _stream())
        _LAST_AMAX[0] = am                   # norm_act() attaches it to the tensor the caller receives
        ctx.cfg = (N, C, HW, int(per_channel), _NORM_MODE[mode], emit)
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        N, C, HW, per_channel, mode, emit = ctx.cfg
        dy = _c(dy)
        _chk(dy)
        G = C if per_channel else N * C
        bsums = torch.empty(_q("objgan_norm_ws_floats", N, C, HW, int(per_channel)), dtype=_F32, device=x.device)
        dx = torch.empty_like(x)
        dgamma = dbeta = None
        if gamma is not None:
            dgamma = torch.empty_like(gamma)
            dbeta = torch.empty_like(beta)
        am = ((_amax_zeroed(x.device) if emit == 2 else torch.empty(_AMAX_SLOTS, dtype=_F32, device=x.device))
              if (emit and _amax_wanted(dx.numel())) else None)
        _lib.call("objgan_norm_backward", _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta),
                  _p(bsums), _p(dx), _p(dgamma), _p(dbeta), N, C, HW, per_channel, mode, _p(am), _stream())
        if am is not None:
            _amax_attach(dx, am)             # (the convolution backward in front of this layer reads dx next)
        dres = dy if ctx.has_res else None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None


def norm_act(x, gamma=None, beta=None, residual=None, running_mean=None, running_var=None,
             per_channel=False, mode=None, eps=1e-5, momentum=0.1):
    """BatchNorm (per_channel=True, batch statistics) or InstanceNorm (per_channel=False)
    fused with GLU / LeakyReLU(0.2) and an optional residual add."""
    _LAST_AMAX[0] = None
    y = _NormActFn.apply(x, gamma, beta, residual, running_mean, running_var, per_channel, mode,
                         eps, momentum)
    if _LAST_AMAX[0] is not None:
        _amax_attach(y, _LAST_AMAX[0])
        _LAST_AMAX[0] = None
    return y


def norm_act_eval(x, gamma, beta, running_mean, running_var, mode=None, eps=1e-5):
    """Eval-mode BatchNorm (running statistics) fused with GLU / LeakyReLU: forward only (generator
    sampling with the EMA weights)."""
    _chk(x, gamma, beta, running_mean, running_var)
    if x.requires_grad and torch.is_grad_enabled():
        raise _lib.ObjganHipError("eval-mode BatchNorm is forward-only: wrap the call in torch.no_grad()")
    x = _c(x)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C) if x.numel() else 0
    Co = C // 2 if mode == "glu" else C
    y = torch.empty((N, Co) + tuple(x.shape[2:]), dtype=_F32, device=x.device)
    rstd = torch.rsqrt(running_var + eps)
    _lib.call("objgan_norm_apply", _p(x), _p(y), None, _p(gamma), _p(beta), _p(running_mean), _p(rstd),
              N, C, HW, 1, _NORM_MODE[mode], _stream())
    return y


# ----------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------
def _mask_u8(mask):
    if mask is None:
        return None
    if not mask.is_cuda:
        raise _lib.ObjganHipError("attention mask must live on the GPU")
    return _c(mask.to(torch.uint8))


class _AttnGeneralFn(torch.autograd.Function):
    """weightedContext, attn = GlobalAttentionGeneral core (reference GlobalAttention.py:95-120)
    given the projected context sourceT [B, idf, L]."""

    @staticmethod
    def forward(ctx, x, src, mask_u8):
        _chk(x, src)
        x = _c(x)
        src = _c(src)
        B, idf = x.shape[0], x.shape[1]
        Q = x.shape[2] * x.shape[3]
        L = src.shape[2]
        wc = torch.empty_like(x)
        attn = torch.empty((B, L, x.shape[2], x.shape[3]), dtype=_F32, device=x.device)
        _lib.call("objgan_attn_general_forward", _p(x), _p(src), _p(mask_u8), _p(wc), _p(attn),
                  B, idf, Q, L, _stream())
        ctx.save_for_backward(x, src, attn)
        return wc, attn

    @staticmethod
    def backward(ctx, dwc, dattn):
        x, src, attn = ctx.saved_tensors
        B, idf = x.shape[0], x.shape[1]
        Q = x.shape[2] * x.shape[3]
        L = src.shape[2]
        dwc = _c(dwc)
        dattn = _c(dattn) if dattn is not None else None
        dx = torch.empty_like(x)
        dsrc = torch.empty_like(src)                  # fully written by the ordered combine
        ws = torch.empty(_lib.load().objgan_attn_general_backward_ws_floats(B, idf, Q, L), dtype=_F32, device=x.device)
        _lib.call("objgan_attn_general_backward", _p(x), _p(src), _p(attn), _p(dwc), _p(dattn),
                  _p(dx), _p(dsrc), B, idf, Q, L, _p(ws), _stream())
        return dx, dsrc, None


def attn_general(x, src, mask=None):
    return _AttnGeneralFn.apply(x, src, _mask_u8(mask))


class _AttnBUFn(torch.autograd.Function):
    """GlobalBUAttentionGeneral core (reference GlobalAttention.py:153-179): scores from the
    label features and GloVe words, context from the projected word embeddings `src`."""

    @staticmethod
    def forward(ctx, tgt, ctx1, src, mask_u8, normalize, eps):
        _chk(tgt, ctx1, src)
        if tgt.requires_grad or ctx1.requires_grad:
            raise _lib.ObjganHipError(
                "attn_bu: gradients w.r.t. the label / GloVe features are not implemented "
                "(they are constants on the training hot path)")
        tgt, ctx1, src = _c(tgt), _c(ctx1), _c(src)
        B, d2, R = tgt.shape[0], tgt.shape[1], tgt.shape[2] * tgt.shape[3]
        idf, L = src.shape[1], src.shape[2]
        wc = torch.empty((B, idf, tgt.shape[2], tgt.shape[3]), dtype=_F32, device=tgt.device)
        attn = torch.empty((B, L, tgt.shape[2], tgt.shape[3]), dtype=_F32, device=tgt.device)
        _lib.call("objgan_attn_bu_forward", _p(tgt), _p(ctx1), _p(src), _p(mask_u8), _p(wc), _p(attn),
                  B, d2, idf, R, L, int(normalize), float(eps), _stream())
        ctx.save_for_backward(attn)
        ctx.dims = (B, idf, R, L)
        ctx.mark_non_differentiable(attn)
        return wc, attn

    @staticmethod
    def backward(ctx, dwc, dattn):
        (attn,) = ctx.saved_tensors
        B, idf, R, L = ctx.dims
        dwc = _c(dwc)
        dsrc = torch.empty((B, idf, L), dtype=_F32, device=dwc.device)
        _lib.call("objgan_attn_bu_backward", _p(dwc), _p(attn), _p(dsrc), B, idf, R, L, _stream())
        return None, None, dsrc, None, None, None


def attn_bu(tgt, ctx1, src, mask=None, normalize=True, eps=1e-8):
    return _AttnBUFn.apply(tgt, ctx1, src, _mask_u8(mask), normalize, eps)


class _MaskedMaxFn(torch.autograd.Function):
    """out[b, c, p] = max_r f[b, c, r] * m[b, r, (c,) p]  (reference miscc/utils.py:401-413)."""

    @staticmethod
    def forward(ctx, f, m, ih, iw):
        _chk(f, m)
        B, num, R = f.shape[0], f.shape[1], f.shape[2]
        f2 = _c(f.reshape(B, num, R))
        P = ih * iw
        if m.dim() == 4:                       # [B, R, ih, iw]: one mask for every channel
            if not (m.stride(3) == 1 and m.stride(2) == iw):
                m = _c(m)                      # (a [:, :R] slice keeps its strides: no copy)
            sb, sr, sc = m.stride(0), m.stride(1), 0
        elif m.dim() == 5:                     # [B, R, num, ih, iw] (the reference's repeated form)
            if m.stride(2) == 0 and m.stride(4) == 1 and m.stride(3) == iw:
                sb, sr, sc = m.stride(0), m.stride(1), 0
            else:
                m = _c(m)
                sb, sr, sc = R * num * P, num * P, P
        else:
            raise _lib.ObjganHipError("masked_max: mask must be 4-D or 5-D")
        out = torch.empty((B, num, ih, iw), dtype=_F32, device=f.device)
        _lib.call("objgan_masked_max_forward", _p(f2), _p(m), _p(out), B, num, R, P, sb, sr, sc, _stream())
        ctx.save_for_backward(f2, m)
        ctx.geom = (B, num, R, P, sb, sr, sc, tuple(f.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        f2, m = ctx.saved_tensors
        B, num, R, P, sb, sr, sc, fshape = ctx.geom
        dout = _c(dout)
        df = torch.empty((B, num, R), dtype=_F32, device=f2.device)
        ws = torch.empty(_lib.load().objgan_masked_max_backward_ws_floats(B, num, R, P), dtype=_F32, device=f2.device)
        _lib.call("objgan_masked_max_backward", _p(f2), _p(m), _p(dout), _p(df), B, num, R, P,
                  sb, sr, sc, _p(ws), _stream())
        return df.reshape(fshape), None, None, None


def masked_max(f, m, ih, iw):
    return _MaskedMaxFn.apply(f, m, ih, iw)


def _bmm_raw(A, B):
    """C[b] = A[b] @ B[b] for 3-D fp32 tensors with arbitrary strides (no copies)."""
    Bt, M, K = A.shape
    N = B.shape[2]
    C = torch.empty((Bt, M, N), dtype=_F32, device=A.device)
    _lib.call("objgan_bmm_strided", _p(A), _p(B), _p(C), Bt, M, N, K,
              A.stride(0), A.stride(1), A.stride(2), B.stride(0), B.stride(1), B.stride(2),
              C.stride(0), C.stride(1), C.stride(2), _stream())
    return C


class _BmmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B):
        _chk(A, B)
        if A.dim() != 3 or B.dim() != 3 or A.shape[0] != B.shape[0] or A.shape[2] != B.shape[1]:
            raise _lib.ObjganHipError("bmm: shapes %s x %s" % (tuple(A.shape), tuple(B.shape)))
        ctx.save_for_backward(A, B)
        return _bmm_raw(A, B)

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        _chk(dC)
        dA = _bmm_raw(dC, B.transpose(1, 2)) if ctx.needs_input_grad[0] else None
        dB = _bmm_raw(A.transpose(1, 2), dC) if ctx.needs_input_grad[1] else None
        return dA, dB


def bmm(A, B):
    """torch.bmm for small fp32 batches on one launch (any strides)."""
    return _BmmFn.apply(A, B)


class _BceConstFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, target):
        _chk(prob)
        prob = _c(prob)
        out = torch.empty((), dtype=_F32, device=prob.device)
        _lib.call("objgan_bce_const_forward", _p(prob), _p(out), prob.numel(), float(target), _stream())
        ctx.save_for_backward(prob)
        ctx.target = float(target)
        return out

    @staticmethod
    def backward(ctx, g):
        (prob,) = ctx.saved_tensors
        g = _c(g)
        _chk(g)
        dp = torch.empty_like(prob)
        _lib.call("objgan_bce_const_backward", _p(prob), _p(g), _p(dp), prob.numel(), ctx.target, _stream())
        return dp, None


def bce_const(prob, target):
    """F.binary_cross_entropy(prob, full_like(prob, target)) for target 0 / 1, one launch each way."""
    return _BceConstFn.apply(prob, target)


class _SoftmaxStridedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dim, scale, lens, rowvalid):
        _chk(x)
        x = _c(x)
        d = dim % x.dim()
        outer = int(math.prod(x.shape[:d]))
        n = x.shape[d]
        inner = int(math.prod(x.shape[d + 1:]))
        y = torch.empty_like(x)
        _lib.call("objgan_softmax_strided_forward", _p(x), _p(y), outer, n, inner, float(scale),
                  _p(lens), 0 if lens is None else lens.numel(), _p(rowvalid), _stream())
        ctx.save_for_backward(y)
        ctx.geom = (outer, n, inner, float(scale))
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        outer, n, inner, scale = ctx.geom
        dy = _c(dy)
        dx = torch.empty_like(y)
        _lib.call("objgan_softmax_strided_backward", _p(y), _p(dy), _p(dx), outer, n, inner, scale, _stream())
        return dx, None, None, None, None


def softmax_strided(x, dim, scale=1.0, lens=None, rowvalid=None):
    """softmax(scale * x) along `dim`; `lens` (int32, per outer row modulo len(lens)) truncates the
    softmax span, `rowvalid` (uint8 per outer row) zeroes whole rows."""
    return _SoftmaxStridedFn.apply(x, dim, scale, lens, rowvalid)


# ----------------------------------------------------------------------------------------------
# ROIAlign, pooling, resize
# ----------------------------------------------------------------------------------------------
class RoIAlignFunction(torch.autograd.Function):
    """reference image_generation/models/roi_align/functions/roi_align.py:7-51 (new-style)."""

    @staticmethod
    def forward(ctx, features, rois, aligned_height, aligned_width, spatial_scale):
        _chk(features, rois)
        features = _c(features)
        rois = _c(rois)
        B, C, H, W = features.shape
        num_rois = rois.shape[0]
        roi_cols = rois.shape[1] if rois.dim() == 2 else 0
        out = torch.zeros((num_rois, C, aligned_height, aligned_width), dtype=_F32, device=features.device)
        _lib.call("objgan_roi_align_forward", _p(features), _p(rois), _p(out), num_rois, roi_cols,
                  C, H, W, aligned_height, aligned_width, float(spatial_scale), _stream())
        ctx.save_for_backward(rois)
        ctx.geom = (B, C, H, W, aligned_height, aligned_width, float(spatial_scale))
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        B, C, H, W, ah, aw, scale = ctx.geom
        grad_output = _c(grad_output)
        _chk(grad_output)
        grad_input = torch.zeros((B, C, H, W), dtype=_F32, device=grad_output.device)
        need = _lib.load().objgan_roi_align_backward_ws_floats(B, rois.shape[0], C, H, W, ah, aw)
        ws = torch.empty((max(int(need), 1),), dtype=_F32, device=grad_output.device)
        _lib.call("objgan_roi_align_backward_ordered", _p(grad_output), _p(rois), _p(grad_input), B,
                  rois.shape[0], rois.shape[1], C, H, W, ah, aw, scale, _p(ws), int(need), _stream())
        return grad_input, None, None, None, None


def roi_align(features, rois, aligned_height, aligned_width, spatial_scale):
    return RoIAlignFunction.apply(features, rois, int(aligned_height), int(aligned_width), float(spatial_scale))


class _AvgPool2s1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = _c(x)
        ih, iw = x.shape[-2], x.shape[-1]
        planes = x.numel() // (ih * iw) if x.numel() else 0
        y = torch.empty(tuple(x.shape[:-2]) + (ih - 1, iw - 1), dtype=_F32, device=x.device)
        _lib.call("objgan_avgpool2s1_forward", _p(x), _p(y), planes, ih, iw, _stream())
        ctx.geom = (tuple(x.shape), planes, ih, iw)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, planes, ih, iw = ctx.geom
        dy = _c(dy)
        dx = torch.empty(shape, dtype=_F32, device=dy.device)
        _lib.call("objgan_avgpool2s1_backward", _p(dy), _p(dx), planes, ih, iw, _stream())
        return dx


def avgpool2s1(x):
    return _AvgPool2s1Fn.apply(x)


class _Pool2dFn(torch.autograd.Function):
    """F.max_pool2d(x, k, s) / F.avg_pool2d(x, k, s, p) of the frozen Inception-v3 encoder."""

    @staticmethod
    def forward(ctx, x, k, s, p, mode):
        _chk(x)
        x = _c(x)
        H, W = x.shape[-2], x.shape[-1]
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        planes = x.numel() // (H * W) if x.numel() else 0
        y = torch.empty(tuple(x.shape[:-2]) + (OH, OW), dtype=_F32, device=x.device)
        need_idx = mode == 0 and ctx.needs_input_grad[0]
        idx = torch.empty(y.shape, dtype=torch.int32, device=x.device) if need_idx else None
        _lib.call("objgan_pool2d_forward", _p(x), _p(y), _p(idx), planes, H, W, OH, OW, k, s, p, mode, _stream())
        ctx.geom = (tuple(x.shape), planes, H, W, OH, OW, k, s, p, mode)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, planes, H, W, OH, OW, k, s, p, mode = ctx.geom
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty(shape, dtype=_F32, device=dy.device)
        _lib.call("objgan_pool2d_backward", _p(dy), _p(idx), _p(dx), planes, H, W, OH, OW, k, s, p, mode, _stream())
        return dx, None, None, None, None


def max_pool2d(x, kernel_size, stride):
    return _Pool2dFn.apply(x, int(kernel_size), int(stride), 0, 0)


def avg_pool2d(x, kernel_size, stride=None, padding=0):
    k = int(kernel_size)
    return _Pool2dFn.apply(x, k, k if stride is None else int(stride), int(padding), 1)


class _BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        _chk(x)
        x = _c(x)
        ih, iw = x.shape[-2], x.shape[-1]
        planes = x.numel() // (ih * iw) if x.numel() else 0
        y = torch.empty(tuple(x.shape[:-2]) + (oh, ow), dtype=_F32, device=x.device)
        _lib.call("objgan_bilinear_forward", _p(x), _p(y), planes, ih, iw, oh, ow, _stream())
        ctx.geom = (tuple(x.shape), planes, ih, iw, oh, ow)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, planes, ih, iw, oh, ow = ctx.geom
        dy = _c(dy)
        dx = torch.empty(shape, dtype=_F32, device=dy.device)
        _lib.call("objgan_bilinear_backward", _p(dy), _p(dx), planes, ih, iw, oh, ow, _stream())
        return dx, None, None


def bilinear_resize(x, oh, ow):
    """F.interpolate(x, size=(oh, ow), mode='bilinear', align_corners=True)."""
    return _BilinearFn.apply(x, int(oh), int(ow))


# ---- training images: baseline JPEG decode on the device (csrc/jpeg.hip) ---------------------------------
class JpegUnsupported(_lib.ObjganHipError):
    """the file is not a baseline / sequential Huffman JPEG in a supported sampling (progressive, arithmetic, CMYK ...):
    the caller routes it to its host decoder knowingly -- nothing is decoded approximately on the device"""

    REASONS = {1: "not a JPEG file", 2: "progressive / lossless / arithmetic-coded frame", 3: "sample precision is not 8 bits",
               4: "neither 1 nor 3 components", 5: "unsupported chroma sampling", 6: "non-interleaved or missing scan",
               7: "missing / oversized table", 8: "truncated file", 9: "RGB-coded (Adobe transform 0 / 'RGB' component ids)"}

    def __init__(self, index, reason):
        super().__init__("JPEG %d: %s" % (index, self.REASONS.get(reason, "reason %d" % reason)))
        self.index, self.reason = index, reason


def jpeg_parse(files):
    """files: the JPEG files of a batch (bytes-like).  HOST ONLY (no GPU call): -> (descriptors uint8 [n, desc bytes] as a
    numpy array, heads int32 [n, 10] view: width, height, components, hmax, vmax, mcux, mcuy, restart interval, reason,
    scan offset).  A file the device path does not decode has heads[i, 8] != 0 (see JpegUnsupported.REASONS)."""
    import numpy as np
    lib = _lib.load()
    nb = int(lib.objgan_jpeg_desc_bytes())
    descs = np.zeros((len(files), nb), np.uint8)
    for i, f in enumerate(files):
        buf = bytes(f) if not isinstance(f, bytes) else f
        lib.objgan_jpeg_parse(buf, len(buf), descs[i].ctypes.data_as(ctypes.c_void_p))
    heads = descs[:, :40].view(np.int32) if len(files) else np.zeros((0, 10), np.int32)
    return descs, heads


class JpegIndexCache(object):
    """Entropy indexes of the files of a data set, kept ON THE DEVICE (48 bytes per MCU row: 3 KB for a 640 x 480 image,
    ~250 MB for all of COCO train -- nothing next to 288 GB): key -> uint8 tensor.  The first decode of a file walks its
    Huffman scan with one lane and leaves the index; every later decode (the next epoch) runs one lane per MCU row."""

    def __init__(self):
        self.entries = {}
        self.hits = self.misses = 0

    def get(self, key, nbytes, rows, seg_bytes):
        e = self.entries.get(key)
        if e is not None and e[0] == nbytes and e[1].numel() == rows * seg_bytes:
            self.hits += 1
            return e[1]
        self.misses += 1
        return None

    def put(self, key, nbytes, index):
        self.entries[key] = (nbytes, index)


def jpeg_decode_batch(files, device, cache=None, keys=None):
    """files: baseline JPEG files (bytes-like) -> (rgb, offs, hs, ws): the decoded images back to back in ONE uint8 device
    buffer (image i: [hs[i], ws[i], 3] at byte offs[i]) -- bit for bit PIL.Image.open(f).convert('RGB') (reference
    miscc/load.py:141-151).  Only the file bytes cross PCIe.  Raises JpegUnsupported (with the index) for a file that is
    not baseline.  cache / keys (a JpegIndexCache and one hashable key per file): files whose entropy index is cached are
    decoded by one lane per MCU row, the others by one lane -- and leave their index in the cache."""
    import numpy as np
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.ObjganHipError("jpeg_decode_batch: needs a HIP device (no CPU path)")
    files = [bytes(f) if not isinstance(f, bytes) else f for f in files]
    n = len(files)
    descs, heads = jpeg_parse(files)
    for i in range(n):
        if heads[i, 8] != 0:
            raise JpegUnsupported(i, int(heads[i, 8]))
    lib = _lib.load()
    seg_bytes = int(lib.objgan_jpeg_seg_bytes())
    hs, ws = [int(h) for h in heads[:, 1]], [int(w) for w in heads[:, 0]]
    rows = [int(r) for r in heads[:, 6]]
    sizes = [(len(f) + 15) // 16 * 16 for f in files]                 # 16-byte aligned file starts (vector loads of the window)
    foffs = np.concatenate(([0], np.cumsum(sizes)[:-1])).astype(np.int64)
    osz = [(h * w * 3 + 15) // 16 * 16 for h, w in zip(hs, ws)]
    ooffs = np.concatenate(([0], np.cumsum(osz)[:-1])).astype(np.int64)
    have = [None] * n
    if cache is not None and keys is not None:
        have = [cache.get(k, len(f), r, seg_bytes) for k, f, r in zip(keys, files, rows)]
    nsegs = np.asarray([0 if h is None else r for h, r in zip(have, rows)], np.int32)
    ws_bytes = int(lib.objgan_jpeg_plan(descs.ctypes.data_as(ctypes.c_void_p), n,
                                        foffs.ctypes.data_as(ctypes.c_void_p), ooffs.ctypes.data_as(ctypes.c_void_p),
                                        nsegs.ctypes.data_as(ctypes.c_void_p)))
    if ws_bytes <= 0:
        raise _lib.ObjganHipError("objgan_jpeg_plan: bad descriptor")
    host = torch.empty(int(sum(sizes)) + descs.size, dtype=torch.uint8).pin_memory()
    flat = host.numpy()
    for f, o in zip(files, foffs):
        flat[o:o + len(f)] = np.frombuffer(f, np.uint8)
    d0 = int(sum(sizes))
    flat[d0:] = descs.reshape(-1)
    dev = host.to(device, non_blocking=True)                          # files + descriptors: one upload
    out = torch.empty(int(sum(osz)), dtype=torch.uint8, device=device)
    work = torch.empty((ws_bytes + 15) // 16 * 16, dtype=torch.uint8, device=device)
    idx_in = torch.cat([h for h in have if h is not None]) if any(h is not None for h in have) else None
    want_out = cache is not None and keys is not None and any(h is None for h in have)
    idx_out = torch.empty(sum(rows) * seg_bytes, dtype=torch.uint8, device=device) if want_out else None
    _lib.call("objgan_jpeg_decode", _p(dev), descs.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dev.data_ptr() + d0), n,
              _p(out), _p(work), ws_bytes, _p(idx_in), _p(idx_out), _stream())
    if want_out:
        o = 0
        for k, f, r, h in zip(keys, files, rows, have):
            if h is None:
                cache.put(k, len(f), idx_out[o * seg_bytes:(o + r) * seg_bytes])
            o += r
    return out, ooffs, hs, ws


def images_to_device(items, device, cache=None, keys=None):
    """The `device_jpeg` hand-over of a batch: every item is either a JPEG file (1-D uint8 tensor / bytes: decoded on the
    device) or an image the loader had to decode on the host (uint8 [H, W, 3]: a progressive / CMYK file the device path
    refuses, uploaded as it is) -> (buffer, offs, hs, ws) in the items' order, the form resize_pil_bilinear_device reads."""
    import numpy as np
    device = torch.device(device)
    kinds = []
    for it in items:
        a = it.numpy() if torch.is_tensor(it) else (np.frombuffer(it, np.uint8) if isinstance(it, (bytes, bytearray)) else np.asarray(it))
        kinds.append(a)
    jidx = [i for i, a in enumerate(kinds) if a.ndim == 1]
    hidx = [i for i, a in enumerate(kinds) if a.ndim == 3]
    if len(jidx) + len(hidx) != len(kinds):
        raise _lib.ObjganHipError("images_to_device: items must be JPEG byte strings (1-D uint8) or uint8 [H, W, 3] images")
    jkeys = None if keys is None else [keys[i] for i in jidx]
    if not hidx:
        return jpeg_decode_batch([kinds[i].tobytes() for i in jidx], device, cache, jkeys)
    parts, offs, hs, ws = [], [0] * len(kinds), [0] * len(kinds), [0] * len(kinds)
    total = 0
    if jidx:
        out, jo, jh, jw = jpeg_decode_batch([kinds[i].tobytes() for i in jidx], device, cache, jkeys)
        parts.append(out)
        for k, i in enumerate(jidx):
            offs[i], hs[i], ws[i] = int(jo[k]), jh[k], jw[k]
        total = out.numel()
    host = torch.empty(sum((kinds[i].size + 15) // 16 * 16 for i in hidx), dtype=torch.uint8).pin_memory()
    o = 0
    for i in hidx:
        a = np.ascontiguousarray(kinds[i], dtype=np.uint8)
        if a.shape[2] != 3:
            raise _lib.ObjganHipError("images_to_device: host-decoded images must be uint8 [H, W, 3], got %s" % (a.shape,))
        host.numpy()[o:o + a.size] = a.reshape(-1)
        offs[i], hs[i], ws[i] = total + o, a.shape[0], a.shape[1]
        o += (a.size + 15) // 16 * 16
    parts.append(host.to(device, non_blocking=True))
    return torch.cat(parts), np.asarray(offs, np.int64), hs, ws


def jpeg_decode(files, device, cache=None, keys=None):
    """-> list of uint8 [H, W, 3] device tensors (views of one buffer)"""
    out, offs, hs, ws = jpeg_decode_batch(files, device, cache, keys)
    return [out[int(o):int(o) + h * w * 3].view(h, w, 3) for o, h, w in zip(offs, hs, ws)]


# ---- training images: Pillow's antialiased bilinear resize + ToTensor + Normalize on the device -------
def resize_pil_bilinear_device(src, offs, hs, ws, sizes):
    """resize_pil_bilinear on images that are ALREADY on the device (jpeg_decode_batch's hand-over): src uint8 buffer, image b
    [hs[b], ws[b], 3] at byte offs[b]."""
    import numpy as np
    device = src.device
    B = len(hs)
    hs_d = torch.tensor(list(hs), dtype=torch.int32).to(device)
    ws_d = torch.tensor(list(ws), dtype=torch.int32).to(device)
    offs_d = torch.from_numpy(np.asarray(offs, np.int64)).to(device)
    Hmax = max(hs)
    side = max(max(hs), max(ws))
    outs = []
    for S in sizes:
        S = int(S)
        kmax = int(_lib.load().objgan_resize_pil_kmax(side, S))
        coef = torch.empty(B * 2 * S * (kmax + 2), dtype=torch.int32, device=device)
        tmp = torch.empty(B * Hmax * S * 3, dtype=torch.uint8, device=device)
        out = torch.empty((B, 3, S, S), dtype=_F32, device=device)
        _lib.call("objgan_resize_pil_rgb8", _p(src), _p(offs_d), _p(hs_d), _p(ws_d), B, Hmax, kmax, S,
                  _p(coef), _p(tmp), _p(out), _stream())
        outs.append(out)
    return outs


def resize_pil_bilinear(images, sizes, device):
    """images: the decoded RGB images of a batch as uint8 [H, W, 3] host tensors / arrays (any sizes);
    -> [B, 3, S, S] float32 on `device` for every S in `sizes`, bit for bit
    Normalize(.5, .5)(ToTensor(transforms.Resize((S, S))(img))) of reference miscc/load.py:141-150.
    The images cross PCIe once, as bytes (csrc/resize_pil.hip)."""
    import numpy as np
    arrs = [np.ascontiguousarray(im.numpy() if torch.is_tensor(im) else im, dtype=np.uint8) for im in images]
    for a in arrs:
        if a.ndim != 3 or a.shape[2] != 3 or a.shape[0] < 1 or a.shape[1] < 1:
            raise _lib.ObjganHipError("resize_pil_bilinear: images must be uint8 [H, W, 3], got %s" % (a.shape,))
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.ObjganHipError("resize_pil_bilinear: needs a HIP device (no CPU path)")
    B = len(arrs)
    sizes_b = [a.size for a in arrs]
    offs = np.concatenate(([0], np.cumsum(sizes_b)[:-1])).astype(np.int64) if B else np.zeros(0, np.int64)
    host = torch.empty(int(sum(sizes_b)), dtype=torch.uint8).pin_memory()
    flat = host.numpy()
    for a, o in zip(arrs, offs):
        flat[o:o + a.size] = a.reshape(-1)
    src = host.to(device, non_blocking=True)
    hs = torch.tensor([a.shape[0] for a in arrs], dtype=torch.int32).to(device)
    ws = torch.tensor([a.shape[1] for a in arrs], dtype=torch.int32).to(device)
    offs_d = torch.from_numpy(offs).to(device)
    Hmax = max(a.shape[0] for a in arrs)
    side = max(max(a.shape[0], a.shape[1]) for a in arrs)
    outs = []
    for S in sizes:
        S = int(S)
        kmax = int(_lib.load().objgan_resize_pil_kmax(side, S))
        coef = torch.empty(B * 2 * S * (kmax + 2), dtype=torch.int32, device=device)
        tmp = torch.empty(B * Hmax * S * 3, dtype=torch.uint8, device=device)
        out = torch.empty((B, 3, S, S), dtype=_F32, device=device)
        _lib.call("objgan_resize_pil_rgb8", _p(src), _p(offs_d), _p(hs), _p(ws), B, Hmax, kmax, S,
                  _p(coef), _p(tmp), _p(out), _stream())
        outs.append(out)
    return outs


# ---- per-box instance masks: skimage.transform.resize (= two scipy.ndimage calls) on the device ----------
def _gaussian_taps(sigma, truncate=4.0):
    """the normalised weights scipy.ndimage.gaussian_filter1d uses for this sigma, computed like scipy computes them
    (float64 numpy, radius = int(truncate * sigma + 0.5))"""
    import numpy as np
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def resize_masks(masks, sizes):
    """masks: float64 [..., n, n] on the GPU (n <= 64; the 64 x 64 instance masks of a batch) -> one float64
    [..., S, S] tensor per S in `sizes` (<= 4), each bit for bit `skimage.transform.resize(mask, [S, S])` as scipy
    evaluates skimage's defaults (reference miscc/load.py:160-176: S = 32, 64, 128, 256); csrc/resize_pil.hip."""
    if not masks.is_cuda or masks.dtype != torch.float64 or masks.dim() < 2 or masks.shape[-1] != masks.shape[-2]:
        raise _lib.ObjganHipError("resize_masks: float64 [..., n, n] on the GPU (got %s %s on %s; no CPU path)"
                                  % (masks.dtype, tuple(masks.shape), masks.device))
    n = int(masks.shape[-1])
    sizes = [int(S) for S in sizes]
    if not 2 <= n <= 64 or not 1 <= len(sizes) <= 4:
        raise _lib.ObjganHipError("resize_masks: n <= 64 and at most four sizes")
    src = _c(masks)
    lead = tuple(src.shape[:-2])
    count = 1
    for d in lead:
        count *= int(d)
    outs = [torch.empty(lead + (S, S), dtype=torch.float64, device=src.device) for S in sizes]
    ntaps, taps = [], [0.0] * (17 * len(sizes))
    for k, S in enumerate(sizes):
        if S < n:
            w = _gaussian_taps((n / float(S) - 1.0) / 2.0)
            if len(w) > 17:
                raise _lib.ObjganHipError("resize_masks: shrinking %d -> %d is outside the table (radius <= 8)" % (n, S))
            ntaps.append(len(w))
            taps[17 * k:17 * k + len(w)] = [float(v) for v in w]
        else:
            ntaps.append(0)
    outp = (ctypes.c_void_p * len(sizes))(*[o.data_ptr() for o in outs])
    _lib.call("objgan_mask_resize", _p(src), count, n, len(sizes), _iarr(sizes), outp, _iarr(ntaps),
              (ctypes.c_double * len(taps))(*taps), _stream())
    return outs


# ---- layout-map stem of the object discriminators, evaluated below the 512x512 lift -------------------
# conv3x3(reflect_pad(lift(seg)))[co] = sum_taps (shift_tap o reflect o lift)(conv1x1(seg; W[:, :, tap])[co]):
# the channel contraction is a 1x1 convolution at the source resolution (MFMA kernel), the pixel operators
# are separable per axis and table-driven (csrc/lift_stem.hip).
_LIFT_TABLES = {}


def _lift_axis_tables(n_in, n_out, device):
    """Per-axis operator A_d = shift_d o reflect o bilinear(align_corners) for d = -1, 0, +1: forward tables
    (i0, i1, l1) [3][n_out] and their CSR transposes (off [3][n_in + 1], idx, wt), in the fp32 arithmetic of
    bilinear_fwd_kernel (src = scale * dst)."""
    key = (n_in, n_out, str(device))
    ent = _LIFT_TABLES.get(key)
    if ent is not None:
        return ent
    import numpy as np
    scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
    i0s, i1s, l1s, offs, idxs, wts = [], [], [], [], [], []
    base = 0
    for d in (-1, 0, 1):
        r = np.arange(n_out, dtype=np.int64) + d
        r = np.where(r < 0, -r, r)
        r = np.where(r >= n_out, 2 * (n_out - 1) - r, r)
        src = (scale * r.astype(np.float32)).astype(np.float32)
        i0 = np.minimum(src.astype(np.int32), n_in - 1)
        i1 = i0 + (i0 < n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        l0 = (np.float32(1.0) - l1).astype(np.float32)
        i0s.append(i0); i1s.append(i1); l1s.append(l1)
        q = np.concatenate([i0, i1]).astype(np.int64)
        o = np.concatenate([np.arange(n_out), np.arange(n_out)])
        wv = np.concatenate([l0, l1])
        order = np.lexsort((o, q))
        q, o, wv = q[order], o[order], wv[order]
        off = np.zeros(n_in + 1, np.int64)
        np.add.at(off, q + 1, 1)
        off = np.cumsum(off) + base
        base = int(off[-1])
        offs.append(off); idxs.append(o); wts.append(wv)
    def t(arrays, dt, flat=False):
        a = np.concatenate(arrays) if flat else np.stack(arrays)
        return torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(device)
    longest = max(int((o[1:] - o[:-1]).max()) for o in offs)
    if longest > 8:      # LIFT_MAXE of csrc/lift_stem.hip: entries a kernel thread keeps in registers
        raise _lib.ObjganHipError("lift_stem_conv: lift factor %d -> %d too large for the adjoint kernel" % (n_in, n_out))
    ent = (t(i0s, np.int32), t(i1s, np.int32), t(l1s, np.float32),
           t(offs, np.int32), t(idxs, np.int32, True), t(wts, np.float32, True))
    _LIFT_TABLES[key] = ent
    return ent


class _LiftTapsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, bias, SH, SW):
        _chk(z, bias)
        z = _c(z)
        N, C9, h, w = z.shape
        Mo = C9 // 9
        rows = _lift_axis_tables(h, SH, z.device)
        cols = _lift_axis_tables(w, SW, z.device)
        y = torch.empty((N, Mo, SH, SW), dtype=_F32, device=z.device)
        scratch = torch.empty(N * 3 * Mo * h * SW, dtype=_F32, device=z.device)
        _lib.call("objgan_lift_taps_forward", _p(z), _p(bias), _p(y), _p(scratch), N, Mo, h, w, SH, SW,
                  _p(cols[0]), _p(cols[1]), _p(cols[2]), _p(rows[0]), _p(rows[1]), _p(rows[2]), _stream())
        ctx.geom = (N, Mo, h, w, SH, SW)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Mo, h, w, SH, SW = ctx.geom
        dy = _c(dy)
        _chk(dy)
        dz = db = None
        if ctx.needs_input_grad[0]:
            rows = _lift_axis_tables(h, SH, dy.device)
            cols = _lift_axis_tables(w, SW, dy.device)
            dz = torch.empty((N, 9 * Mo, h, w), dtype=_F32, device=dy.device)
            scratch = torch.empty(N * 3 * Mo * h * SW, dtype=_F32, device=dy.device)
            _lib.call("objgan_lift_taps_backward", _p(dy), _p(dz), _p(scratch), N, Mo, h, w, SH, SW,
                      _p(rows[3]), _p(rows[4]), _p(rows[5]), _p(cols[3]), _p(cols[4]), _p(cols[5]), _stream())
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = torch.empty(Mo, dtype=_F32, device=dy.device)
            _channel_sum(dy, db, N, Mo, SH * SW)
        return dz, db, None, None


def lift_stem_conv(seg, w, bias, size):
    """conv2d(reflect_pad(F.interpolate(seg, (size, size), 'bilinear', align_corners=True)), w, bias) for a 3x3
    bank w [Mo, C, 3, 3] (reference model.py:1217-1226), without materialising the lifted map."""
    Mo, C = w.shape[0], w.shape[1]
    if w.shape[2] != 3 or w.shape[3] != 3:
        raise _lib.ObjganHipError("lift_stem_conv: 3x3 filter bank expected")
    w9 = w.permute(2, 3, 0, 1).reshape(9 * Mo, C, 1, 1).contiguous()          # row (dh*3 + dw)*Mo + co
    w9._og_nocache = True        # a per-call temporary: caching its bank would pin one dead entry per call
    z = conv2d(seg, w9)
    return _LiftTapsFn.apply(z, bias, int(size), int(size))


# ----------------------------------------------------------------------------------------------
# frozen text encoder
# ----------------------------------------------------------------------------------------------
def lstm_bidir_forward(table, captions, lens, wt_ih, wt_hh, b_ih, b_hh, max_len):
    """Embedding + bidirectional LSTM over packed captions -> (words_emb [B, 2H, max_len],
    sent_emb [B, 2H]); forward only (the encoder is frozen on the training path)."""
    _chk(table, wt_ih, wt_hh, b_ih, b_hh)
    if not captions.is_cuda or captions.dtype != torch.int64:
        raise _lib.ObjganHipError("lstm: captions must be an int64 tensor on the GPU")
    captions = _c(captions)
    lens = _c(lens.to(device=captions.device, dtype=torch.int32))
    B, L = captions.shape
    I, G = wt_ih.shape[1], wt_ih.shape[2]
    H = G // 4
    out = torch.empty((B, 2 * H, int(max_len)), dtype=_F32, device=captions.device)
    hn = torch.empty((B, 2 * H), dtype=_F32, device=captions.device)
    _lib.call("objgan_lstm_bidir_forward", _p(table), _p(captions), _p(lens), _p(wt_ih), _p(wt_hh),
              _p(b_ih), _p(b_hh), _p(out), _p(hn), B, L, int(max_len), I, H, table.shape[0], _stream())
    return out, hn


# ----------------------------------------------------------------------------------------------
# optimiser on flat arenas
# ----------------------------------------------------------------------------------------------
def adam_step_(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, n=None):
    """In-place Adam over flat arenas (first n elements; g may carry trailing bookkeeping slots)."""
    _chk(p, g, m, v)
    n = p.numel() if n is None else int(n)
    _lib.call("objgan_adam_step", _p(p), _p(g), _p(m), _p(v), n, float(lr), float(beta1),
              float(beta2), float(eps), int(step), float(grad_scale), _stream())


def adam_step_gated_(p, g, m, v, lr, beta1, beta2, eps, state, flag, coef, grad_scale=1.0, n=None):
    """Adam taken only if the device scalar `flag` is positive; `state` (3 float64 on the device:
    steps, beta1^steps, beta2^steps) is advanced by the kernel, `coef` is 3 floats of scratch."""
    _chk(p, g, m, v, flag, coef)
    if not state.is_cuda or state.dtype != torch.float64 or state.numel() < 3:
        raise _lib.ObjganHipError("adam_step_gated_: state must be 3 float64 values on the GPU")
    n = p.numel() if n is None else int(n)
    _lib.call("objgan_adam_step_gated", _p(p), _p(g), _p(m), _p(v), n, float(lr), float(beta1),
              float(beta2), float(eps), _p(state), _p(flag), _p(coef), float(grad_scale), _stream())


def ema_update_(avg, p, decay):
    _chk(avg, p)
    _lib.call("objgan_ema_update", _p(avg), _p(p), p.numel(), float(decay), float(1.0 - decay), _stream())
