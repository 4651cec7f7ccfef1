"""Where ``BBBConv2d(fused_conv="auto")`` -- the default -- takes the fused kernels (csrc/conv_lrt*.hip) instead of the stock
sequence of ``bbb_layers.py:146-154`` (two MIOpen convolutions + fused element-wise passes).

The rule (VERDICT r4 #3 / ADVICE r4): a geometry is fused by default only if a DEVICE measurement of the kernels in this tree
shows them at least as fast as the reference's op sequence on the same MI355X -- ``bde_conv_lrt_supported`` only says that a
tiling exists, not that it wins.  The measurements live in ``conv_profit.json`` beside this file, written by
``tools/conv_lrt_bench.py --table`` on the GPU box (one record per layer geometry: forward-only and forward + backward
speed-ups over the reference's sequence, the kernel ABI version they were taken with, the profile they are filed under);
``tools/conv_autotune.py`` writes the same records after timing EVERY candidate tiling of each pass and adds the winners
(``tilings``), which ``apply_tilings`` pins through the library's tuning hooks when the layer is first used.
Records of another ABI version are ignored: a kernel rewritten since its measurement is unmeasured again.

No record -> stock path.  ``fused_conv=True`` forces the fused kernels wherever they have a tiling (tests, benchmarks),
``fused_conv=False`` never uses them.
"""
import json
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_profit.json")
_MIN_GAIN = 1.0          # fused must be at least this many times the stock sequence's speed
_table = None


def _key(c, o, k, stride, padding, h, w):
    return f"C{int(c)}_O{int(o)}_k{int(k)}_s{int(stride)}_p{int(padding)}_{int(h)}x{int(w)}"


def load(path: str = None) -> dict:
    """{"abi": int, "source": str, "layers": {key: {"batch": n, "fwd": x, "fwd_bwd": x}}}; empty when nothing was measured."""
    global _table
    if path is None and _table is not None:
        return _table
    try:
        with open(path or _PATH) as f:
            t = json.load(f)
        if not isinstance(t.get("layers"), dict):
            t = {"abi": 0, "source": "", "layers": {}}
    except (OSError, ValueError):
        t = {"abi": 0, "source": "", "layers": {}}
    if path is None:
        _table = t
    return t


def profitable(x_shape, w_shape, stride, padding, needs_grad: bool, abi: int, table: dict = None) -> bool:
    """True iff the table holds a record for exactly this layer geometry, taken with kernel ABI ``abi`` at a batch size not
    above four times this call's, whose speed-up for the pass at hand (forward only when no gradient will be asked for,
    forward + backward otherwise) is >= 1."""
    t = load() if table is None else table
    if int(t.get("abi", 0)) != int(abi) or stride[0] != stride[1] or padding[0] != padding[1] or w_shape[2] != w_shape[3]:
        return False
    rec = t["layers"].get(_key(x_shape[1], w_shape[0], w_shape[2], stride[0], padding[0], x_shape[2], x_shape[3]))
    if rec is None or int(x_shape[0]) * 4 < int(rec.get("batch", 1)):
        return False
    return float(rec.get("fwd_bwd" if needs_grad else "fwd", 0.0)) >= _MIN_GAIN


_applied = set()


def apply_tilings(ops, x_shape, w_shape, stride, padding, abi: int, table: dict = None) -> bool:
    """Pin the tilings tools/conv_autotune.py recorded for this layer (forward, input gradient, weight gradient), once per
    library handle and layer.  They are per LAUNCH geometry, batch size included: a record taken at another batch leaves the
    planners' own choice in place.  A recorded tiling the current planner no longer offers is skipped."""
    t = load() if table is None else table
    if int(t.get("abi", 0)) != int(abi) or stride[0] != stride[1] or padding[0] != padding[1] or w_shape[2] != w_shape[3]:
        return False
    key = _key(x_shape[1], w_shape[0], w_shape[2], stride[0], padding[0], x_shape[2], x_shape[3])
    rec = t["layers"].get(key)
    if rec is None or int(rec.get("batch", -1)) != int(x_shape[0]) or not rec.get("tilings"):
        return False
    tag = (id(getattr(ops, "lib", ops)), key, int(x_shape[0]))
    if tag in _applied:
        return True
    from .ops import BdeKernelError
    til = rec["tilings"]
    for row in til.get("launch", []):
        try:
            ops.conv_lrt_set_tiling(row[:15], row[15:19])
        except (BdeKernelError, ValueError, IndexError):
            pass
    if til.get("wgrad"):
        try:
            ops.conv_lrt_wgrad_set_tiling(tuple(x_shape), tuple(w_shape), stride, padding, til["wgrad"])
        except (BdeKernelError, ValueError, IndexError):
            pass
    _applied.add(tag)
    return True

