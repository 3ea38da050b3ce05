#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Checks that hipify-perl's output differs from the reference source ONLY by CUDA->HIP API
identifiers and the runtime header name: every line of the translation, with `hip` spelled back to `cuda`, must be
the reference's line (plus the one `#include "hip/hip_runtime.h"` hipify-perl puts in front of a .cu file).

    check_translation.py <reference file> <translated file>
"""
import re
import sys


def back(line):
    line = re.sub(r'[<"]hip/hip_runtime\.h[>"]', "<RUNTIME>", line)
    line = re.sub(r"\bhip(?=[A-Z])", "cuda", line)
    return line


def fwd(line):
    line = re.sub(r"<cuda(_runtime)?\.h>", "<RUNTIME>", line)
    return line


def main():
    ref = open(sys.argv[1]).read().split("\n")
    out = open(sys.argv[2]).read().split("\n")
    if out and out[0].strip() == '#include "hip/hip_runtime.h"' and len(out) == len(ref) + 1:
        out = out[1:]
    if len(ref) != len(out):
        sys.exit(f"line count differs: {len(ref)} vs {len(out)}")
    changed = 0
    for i, (a, b) in enumerate(zip(ref, out), 1):
        if a == b:
            continue
        changed += 1
        if fwd(a) != back(b):
            sys.exit(f"line {i} differs by more than an API identifier:\n  ref: {a}\n  out: {b}")
    print(f"{sys.argv[2]}: {changed} of {len(ref)} lines changed, all of them cuda*->hip* identifiers / the runtime header")


if __name__ == "__main__":
    main()
