#!/usr/bin/env python3
"""MaxSim A/B on one box: the shipped library against another build of it (FUSION_AMD_LIB), each in its own process, alternating,
on random unit-norm tokens; parity of the two score planes is checked on a small case.  Usage: python tools/run_maxsim_ab.py other.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1])
child = "import sys; sys.path.insert(0, %r); from tools.bench_kernels import bench_maxsim; bench_maxsim()" % ROOT
for rep in range(2):
    for name, lib in (("shipped", ""), ("other", other)):
        env = dict(os.environ)
        if lib:
            env["FUSION_AMD_LIB"] = lib
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, cwd=ROOT)
        for ln in out.stdout.splitlines():
            print(json.dumps(dict(build=name, rep=rep, **json.loads(ln))), flush=True)
        if out.returncode:
            print(out.stderr[-2000:], file=sys.stderr)
