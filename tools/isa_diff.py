#!/usr/bin/env python
"""Are the kernels of two device-only assembly listings the same instruction streams?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S file.hip -o new.s      (same for the old tree)
    python tools/isa_diff.py old.s new.s [--match conv3x3_hl16_patch_kernel] [--drop-targ 4]

Compares, kernel by kernel, every instruction line (labels, directives and comments dropped, symbol names masked).
Used in round 5 to show that taking the timing experiments out of the trunk kernel and cutting its body into
csrc/patch_*.inc changed nothing the device executes (32 of 32 instantiations identical).  ``--drop-targ n`` removes the
n-th template argument (1-based) from the OLD listing's mangled names - the argument the refactor removed."""
import argparse
import re


def kernels(path, match):
    ks, cur = {}, None
    for l in open(path):
        m = re.match(r'^(_Z\S+):', l)
        if m and match in m.group(1):
            cur = m.group(1)
            ks[cur] = []
            continue
        if cur is None:
            continue
        t = l.strip()
        if t.startswith('s_endpgm'):
            ks[cur].append('s_endpgm')
            cur = None
            continue
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        ks[cur].append(re.sub(r';.*$', '', re.sub(r'_Z\S+', 'SYM', t)).strip())
    return ks


def drop_targ(name, n):
    m = re.match(r'^(_Z\d+\w+?I)((?:L[ib]\d+E)+)(E.*)$', name)
    if not m:
        return name
    args = re.findall(r'L[ib]\d+E', m.group(2))
    del args[n - 1]
    return m.group(1) + ''.join(args) + m.group(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('old')
    ap.add_argument('new')
    ap.add_argument('--match', default='')
    ap.add_argument('--drop-targ', type=int, default=0)
    a = ap.parse_args()
    old, new = kernels(a.old, a.match), kernels(a.new, a.match)
    if a.drop_targ:
        old = {drop_targ(k, a.drop_targ): v for k, v in old.items()}
    same = 0
    for k, v in new.items():
        if old.get(k) == v:
            same += 1
        else:
            print('DIFFERENT' if k in old else 'NEW      ', k[:110], len(v), 'instructions', '(old: %d)' % len(old[k]) if k in old else '')
    for k in old:
        if k not in new:
            print('GONE     ', k[:110])
    print('%d of %d kernels of the new listing have the instruction stream of the old one' % (same, len(new)))


if __name__ == '__main__':
    main()
