#!/usr/bin/env python3
"""ON THE GPU BOX: run the `-m gpu` parity tests of every device-unverified family and record the families that came out green.

    python tools/device_verify.py [--out gpurun_out/device_verified.json] [--log-dir gpurun_out/verify] [family ...]

For each family of beyond_deep_ensembles_amd/device_verified.py (default: all) the tests that carry
`@pytest.mark.device_unverified("<family>", ...)` are collected (tests/gpu_order_probe.py) and run in a CHILD pytest process
of their own -- a GPU fault in one family's kernels ends that child, not the others.  A family whose tests all passed (and at
least one ran) gets a record bound to the sha256 of its sources; the table is written to --out (gpurun_out/ travels back from
the box; the tree on the box does not), to be copied over beyond_deep_ensembles_amd/device_verified.json and committed together
with the logs.  From then on the default paths of that family are on (device_verified.enabled).

The fused convolution kernels are not in this table: their default follows a MEASUREMENT per layer geometry
(conv_profit.json, tools/conv_autotune.py), not a pass/fail record.
"""
import argparse
import datetime
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("families", nargs="*")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "device_verified.json"))
    ap.add_argument("--log-dir", default=os.path.join(ROOT, "gpurun_out", "verify"))
    ap.add_argument("--timeout", type=int, default=1200, help="seconds per family")
    args = ap.parse_args()
    from beyond_deep_ensembles_amd import device_verified as V
    families = args.families or list(V.FAMILIES)
    os.makedirs(args.log_dir, exist_ok=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    probe = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_order_probe.py")], cwd=ROOT, stdout=subprocess.PIPE,
                           check=True, timeout=600)
    tests = json.loads(probe.stdout.decode())["tests"]
    # this process never touches the GPU (it only starts children): the device's name comes from a child too
    who = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.get_device_name(0) if torch.cuda.is_available() else '')"],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    device = (who.stdout.decode().strip().splitlines() or [""])[-1] or None
    if device is None:
        raise SystemExit("tools/device_verify.py needs a GPU: a record is a statement about a device run")
    table = {"families": {}, "note": "written by tools/device_verify.py"}
    summary = {}
    for fam in families:
        ids = [t["id"] for t in tests if fam in t["families"]]
        log = os.path.join(args.log_dir, f"{fam}.log")
        if not ids:
            summary[fam] = "no tests carry this family"
            continue
        # every family's tests also reach the OTHER families they are marked with (a shell test takes all host paths at once):
        # the gates stay as they are -- the tests ask for their paths explicitly
        cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "-m", "gpu", "--junitxml", log + ".xml"] + ids
        with open(log, "w") as fh:
            try:
                rc = subprocess.run(cmd, cwd=ROOT, stdout=fh, stderr=subprocess.STDOUT, timeout=args.timeout).returncode
            except subprocess.TimeoutExpired:
                rc = -1
        tail = open(log).read()[-600:]
        passed = failed = 0
        try:
            import xml.etree.ElementTree as ET
            suite = ET.parse(log + ".xml").getroot()
            suite = suite if suite.tag == "testsuite" else suite.find("testsuite")
            total, bad = int(suite.get("tests", 0)), int(suite.get("failures", 0)) + int(suite.get("errors", 0))
            passed, failed = total - bad - int(suite.get("skipped", 0)), bad
        except Exception:                                            # noqa: BLE001 -- a child that died writes no report
            pass
        if rc == 0 and passed > 0 and failed == 0:
            table["families"][fam] = {"sha256": V.source_hash(fam), "device": device, "tests_passed": passed,
                                      "date": datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M UTC"),
                                      "log": os.path.relpath(log, ROOT), "sources": list(V.FAMILIES[fam][0])}
            summary[fam] = f"verified: {passed} passed"
        else:
            summary[fam] = f"NOT verified: rc={rc}, {passed} passed, {failed} failed -- {tail.splitlines()[-1] if tail else ''}"
        print(f"{fam}: {summary[fam]}", flush=True)
    with open(args.out, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
