"""Kernel resource table from a device-only assembly listing (hipcc ... --cuda-device-only -S): VGPRs, spills, LDS, scratch."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - \.agpr_count", meta)[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void hd::", "")
    print(f"{name:34s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")

# static instruction mix per kernel (packed / scalar FP32 multiply and add, LDS reads, scalar loads)
print()
labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN2hd\w+):", txt, re.M)]
for i, (pos, lab) in enumerate(labels):
    end = txt.find(".Lfunc_end", pos)
    body = txt[pos:end]
    name = subprocess.run(["c++filt", lab], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void hd::", "")
    c = lambda pat: len(re.findall(r"^\s+" + pat, body, re.M))
    print(f"{name:34s} pk_mul {c('v_pk_mul_f32'):5d} pk_add {c('v_pk_add_f32'):5d} mul {c('v_mul_f32'):5d} add {c('v_add_f32'):5d} "
          f"ds_read {c('ds_read'):4d} ds_write {c('ds_write'):4d} s_load {c('s_load'):4d} lines {body.count(chr(10))}")
