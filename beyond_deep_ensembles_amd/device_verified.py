"""Which code that has NOT yet been green on an MI355X a default-constructed object may run.

The rule (the one ``conv_profit.py`` already applies to the fused convolutions): a kernel -- or a native host
path -- written or rewritten while no GPU was reachable is used BY DEFAULT only once its parity tests have been green on a
device for exactly the sources in this tree.  Until then the default is the path whose kernels have (the streaming SVGD
kernels, torch's adds for the loss sum, the per-particle loop of round 3); the new path stays reachable by asking for it
(``SVGDOptimizer(single_launch="two", host_fast_paths=True)``, ``ops.svgd_step_small``, ``rbf(_small=True)``), which is what
its own tests do.

``device_verified.json`` beside this file holds one record per family:

    {"families": {"svgd_small": {"sha256": "<hash of the family's sources>", "device": "...", "log": "profiles/r06_...", ...}}}

written by ``tools/device_verify.py`` ON THE GPU BOX after the family's ``-m gpu`` parity tests passed there.  A record whose
hash differs from the sources at hand is ignored: code edited since its device run is unverified again.  Both paths of
every gate are HIP kernels of this library -- nothing here selects a CPU or torch fallback for the arithmetic.

``BDE_UNVERIFIED=all`` (or a comma-separated list of families) in the environment switches the named gates on without a
record: for A/B runs on a device (``tools/gpu_r6a.sh``), never set by the product.
"""
import hashlib
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "device_verified.json")

# family -> (sources whose content the record is bound to, what the gate switches, the -m gpu tests that verify it)
FAMILIES = {
    "svgd_small": (
        ("csrc/svgd_small.hip", "csrc/svgd_shared.hpp", "csrc/svgd_gram.hpp", "csrc/bde_common.hpp"),
        "svgd_step_small_kernel (M <= 8, D <= 524,288: the whole update in two launches) as the default of "
        "SVGDOptimizer(single_launch=None) and rbf(); otherwise the streaming kernels (bde_svgd_step streams at every size)",
        "small_model or [small]"),
    "small_step_host": (
        ("csrc/host.cpp", "csrc/svgd_small.hip"),
        "host.cpp small_step_sgd / small_step_adam (both C-ABI calls of the small-model step from one native call) instead "
        "of the two Python wrappers",
        "small_model or [small]"),
    "mean_scalars": (
        ("csrc/svgd.hip", "csrc/bde_common.hpp"),
        "sum_scalars_kernel / host.cpp mean_losses for the returned loss (svgd.py:66,72,105) instead of torch's adds",
        "r5_sum_scalars or [small]"),
    "fast_loop": (
        ("csrc/host.cpp",),
        "SVGDOptimizer._step_fast: ParticleSet.end_begin (end of particle i + begin of particle i + 1 in one native call) "
        "instead of round 3's begin / end calls",
        "[small] or fast_loop"),
}

_table = None
_hashes = {}


def source_hash(family: str) -> str:
    """sha256 over the family's source files (names and bytes, in the order of FAMILIES); "" when one is missing."""
    if family not in _hashes:
        h = hashlib.sha256()
        try:
            for rel in FAMILIES[family][0]:
                with open(os.path.join(_HERE, rel), "rb") as f:
                    h.update(rel.encode() + b"\0" + f.read() + b"\0")
            _hashes[family] = h.hexdigest()
        except OSError:
            _hashes[family] = ""
    return _hashes[family]


def load(path: str = None) -> dict:
    global _table
    if path is None and _table is not None:
        return _table
    try:
        with open(path or _PATH) as f:
            t = json.load(f)
        if not isinstance(t.get("families"), dict):
            t = {"families": {}}
    except (OSError, ValueError):
        t = {"families": {}}
    if path is None:
        _table = t
    return t


def _forced() -> set:
    text = os.environ.get("BDE_UNVERIFIED", "")
    names = {s.strip() for s in text.split(",") if s.strip()}
    return set(FAMILIES) if "all" in names else names


def enabled(family: str, table: dict = None) -> bool:
    """May a DEFAULT path use ``family``?  True iff its record matches the sources at hand (or BDE_UNVERIFIED names it)."""
    if family not in FAMILIES:
        raise KeyError(family)
    if family in _forced():
        return True
    rec = (load() if table is None else table)["families"].get(family)
    if not isinstance(rec, dict):
        return False
    sha = source_hash(family)
    return bool(sha) and rec.get("sha256") == sha


def record(family: str, **fields) -> dict:
    """Write / replace the record of ``family`` for the sources at hand (tools/device_verify.py, on the GPU box)."""
    global _table
    t = load(_PATH)
    t["families"][family] = dict(fields, sha256=source_hash(family))
    with open(_PATH, "w") as f:
        json.dump(t, f, indent=1, sort_keys=True)
        f.write("\n")
    _table = None
    return t["families"][family]


def status() -> dict:
    """{family: "verified" | "unverified (no record)" | "unverified (sources changed since the record)" | "forced"}"""
    out = {}
    fam = load()["families"]
    for name in FAMILIES:
        if name in _forced():
            out[name] = "forced (BDE_UNVERIFIED)"
        elif name not in fam:
            out[name] = "unverified (no record)"
        elif fam[name].get("sha256") != source_hash(name):
            out[name] = "unverified (sources changed since the record)"
        else:
            out[name] = "verified"
    return out


if __name__ == "__main__":                       # python -m beyond_deep_ensembles_amd.device_verified
    for _name, _state in status().items():
        print(f"{_name:16s} {_state:48s} {FAMILIES[_name][1][:110]}")
