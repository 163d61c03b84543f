"""Instruction mix of every loop of one kernel in a hipcc -S listing.  usage: python tools/asm_loops.py file.s kernel_name_substring"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^(\S*%s\S*):" % re.escape(name), s, re.M)
i = m.start()
j = s.index(".Lfunc_end", i) if ".Lfunc_end" in s[i:] else len(s)
lines = s[i:j].split("\n")
labels = {}
for n, l in enumerate(lines):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        labels[mm.group(1)] = n
for n, l in enumerate(lines):
    mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        body = lines[labels[mm.group(1)]:n]
        cnt = lambda pat: sum(1 for x in body if re.search(pat, x))
        valu = cnt(r"^\s+v_") - cnt("v_mfma")
        salu = cnt(r"^\s+s_") - cnt("s_waitcnt") - cnt("s_nop") - cnt("s_barrier")
        other = valu - cnt("v_exp_f32") - cnt("v_cvt_pk_bf16") - cnt(r"v_add_f32") - cnt(r"v_pk_")
        print(f"{mm.group(1)}: {len(body)} lines | mfma {cnt('v_mfma')} exp {cnt('v_exp_f32')} cvt_pk {cnt('v_cvt_pk_bf16')} add {cnt(r'v_add_f32')} "
              f"pk_* {cnt(r'v_pk_')} other-valu {other} | ds_read {cnt('ds_read')} "
              f"ds_write {cnt('ds_write')} buffer_load {cnt('buffer_load')} | s_barrier {cnt('s_barrier')} s_waitcnt {cnt('s_waitcnt')} s_nop {cnt('s_nop')} salu {salu}")
