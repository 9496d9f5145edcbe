#!/usr/bin/env python3
"""Record the md5 of the kernel generator's output for the shipped schedules together with kGeneratorRevision
(tests/golden/generated_source.json).  tests/test_codegen_cpu.py compares: emitted code that changes without a
new revision would let a stale JIT cache entry pass for a current one."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "circuitsimulator_amd", "csrc")


def revision():
    text = open(os.path.join(CSRC, "engine", "codegen.hpp")).read()
    return int(re.search(r"kGeneratorRevision\s*=\s*(\d+)", text).group(1))


def digests():
    out = {}
    gen = os.path.join(CSRC, "build", "csim_codegen")
    for name in ("buffer", "dbmixer"):
        with tempfile.TemporaryDirectory() as d:
            hip = os.path.join(d, "x.hip")
            subprocess.check_call([gen, os.path.join(ROOT, "tests", "golden", name + ".sp"),
                                   os.path.join(CSRC, "schedules", name + ".sched"), hip],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            out[name] = hashlib.md5(open(hip, "rb").read()).hexdigest()
    return out


if __name__ == "__main__":
    rec = {"generator_revision": revision(), "source_md5": digests()}
    path = os.path.join(ROOT, "tests", "golden", "generated_source.json")
    json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, rec)
    sys.exit(0)
