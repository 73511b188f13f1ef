#!/usr/bin/env python3
"""Static instruction count per SOURCE LINE of one kernel (no GPU needed): which statements the instructions of a kernel come from.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only --cuda-device-only --no-gpu-bundle-output -w \
        -c gyeeta_amd/csrc/gys_engine.hip -o /tmp/dev.o && /opt/rocm/lib/llvm/bin/llvm-objdump -d -l /tmp/dev.o > eng.s
  python tools/isa_lines.py k_digest_binsILb0 [second substring of the mangled name]     (reads ./eng.s)
Counts are STATIC: a line inside a loop is counted once.  (Round 2 used it on k_digest_bins; the conclusion drawn from it -- that a
merge's cost is mostly independent of its size -- did not survive measurement, DESIGN.md 10: the kernel is not bound by instruction
issue alone.)"""
import re, sys, collections
kern = sys.argv[1]
cur = None; line = None
cnt = collections.Counter(); valu = collections.Counter()
on = False
for l in open('eng.s'):
    m = re.match(r'^[0-9a-f]+ <(\S+)>:', l)
    if m:
        on = kern in m.group(1) and (len(sys.argv) < 3 or sys.argv[2] in m.group(1))
        continue
    if not on: continue
    m = re.match(r'^; (\S+):(\d+)', l)
    if m:
        line = (m.group(1).split('/')[-1], int(m.group(2))); continue
    m = re.match(r'^\s+(\S+)\s.*//\s*[0-9A-F]+:', l)
    if m and line:
        mn = m.group(1)
        if mn.startswith('s_nop') or mn.startswith('s_code_end'): continue
        cnt[line] += 1
        if mn.startswith('v_'): valu[line] += 1
tot = sum(cnt.values()); print('total', tot, 'valu', sum(valu.values()))
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print('%-22s %5d  all %4d  valu %4d' % (k[0], k[1], v, valu[k]))
