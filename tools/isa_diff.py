#!/usr/bin/env python3
"""Are the product kernels of one translation unit the same instruction stream in two versions of the source?

    python tools/isa_diff.py conv3x3_body16w.hip [--rev HEAD] [--filter body16w_kernel]
Compiles dsen2_amd/csrc/<source> from the working tree and from git revision --rev (with that revision's headers) to
gfx950 ISA with the product's flags (dsen2_amd/build.py) and compares, kernel by kernel, the instruction lists with
local labels normalised.  Used for refactors that must not change the generated code (round 3: removal of the closed
experiments' masks from the bf16 body kernel).  Exit code 1 when a kernel differs or is missing.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsen2_amd import build as b      # noqa: E402


def kernels(path):
    out, name, body = {}, None, []
    for ln in open(path):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            t = ln.strip()
            if t and not t.startswith((';', '.', '//')):
                body.append(re.sub(r'\.L\w+', 'L', re.sub(r'\s+', ' ', t.split(';')[0].strip())))
            if t.startswith('s_endpgm'):
                out[name] = body
                name = None
    return out


def listing(src_dir, source, out):
    flags = [f for f in b.FLAGS if f != '-fPIC']
    subprocess.check_call([b.HIPCC] + flags + ['-S', '--cuda-device-only', os.path.join(src_dir, source), '-o', out],
                          stderr=subprocess.DEVNULL)
    return kernels(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('source')
    ap.add_argument('--rev', default='HEAD')
    ap.add_argument('--filter', default='', help='only kernels whose mangled name contains this')
    args = ap.parse_args()
    with tempfile.TemporaryDirectory(prefix='dsen2_isa_') as tmp:
        old = os.path.join(tmp, 'old', 'dsen2_amd', 'csrc')
        os.makedirs(old)
        os.makedirs(os.path.join(tmp, 'old', 'include'))
        files = subprocess.check_output(['git', 'ls-tree', '--name-only', args.rev, 'dsen2_amd/csrc/', 'include/'], cwd=ROOT, text=True).split()
        for f in files:
            with open(os.path.join(tmp, 'old', f), 'wb') as fh:
                fh.write(subprocess.check_output(['git', 'show', '%s:%s' % (args.rev, f)], cwd=ROOT))
        a = listing(old, args.source, os.path.join(tmp, 'old.s'))
        c = listing(b.CSRC, args.source, os.path.join(tmp, 'new.s'))
    bad = 0
    for k in sorted(a):
        if args.filter not in k:
            continue
        same = a[k] == c.get(k)
        bad += not same
        print('%-110s %5d instructions  %s' % (k[:110], len(a[k]), 'same' if same else ('DIFFERENT' if k in c else 'MISSING')))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
