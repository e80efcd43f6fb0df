"""Markdown table of tools/small_parts.py's output (profiles/r04_small_parts.txt) for DESIGN.md section 5.
usage: python tools/small_parts_table.py profiles/r04_small_parts.txt"""
import json
import sys

rows = [json.loads(ln) for ln in open(sys.argv[1]) if ln.startswith("{")]
print("| part | owned cells (ghosts) | form of the RHS step | kernel (RHS / Euler) | round 3 chain: pack, RCCL, unpack, kernel | direct receive: pack, RCCL, kernel | direct receive + fused pack: RCCL, kernel in order / two-stream form | signalled form (round 4 files only; the form is tools/probes/round4_forms.patch) |")
print("|---|---|---|---|---|---|---|---|")
for d in rows:
    k, ke = d["kernel_rhs"][0], d["kernel_euler"][0]
    f = lambda key, base: f"{d[key][0]:.1f} ({d[key][0] / base:.2f})"
    print(f"| {d['part']} | {d['cells']} ({d['ghosts']}) | {'overlapped' if d['overlapped_form'] else 'in order'} | {k:.1f} / {ke:.1f} | "
          f"RHS {f('r03_rhs', k)}, Euler {f('r03_euler', ke)} | RHS {f('rhs_direct', k)}, Euler {f('euler_direct', ke)} | "
          + (f"Euler {f('euler_fused_in_order', ke)} / {f('euler_fused_overlapped', ke)} | Euler {f('euler_fused_signalled', ke) if 'euler_fused_signalled' in d else 'n/a'} |"
             if "euler_fused_in_order" in d else f"Euler {f('euler_fused', ke)} | |"))
