"""Static per-phase instruction budget of a render kernel.

Build the kernels with -DRT_PHASE_MARKERS -S (see `make -C raytracer_amd/csrc phases`): every
phase of the per-pixel walk then starts with a comment line `; RT_PHASE name` in the assembly.
This script walks one kernel in layout order and attributes every instruction to the phase of
the last marker before it ("fallback" = the plain-operator paths behind the range guards;
"resume" returns to the phase that was current before the fallback).  With --weights it
multiplies by how often a wave enters each phase (a JSON file written by
tools/phase_profile.py from the counting build on the GPU) and prints the estimated executed
vector instructions per launch beside the static counts.

usage: phase_budget.py FILE.s [KERNEL_SUBSTRING] [--weights weights.json]
"""
import collections
import json
import re
import sys

argv = sys.argv[1:]
weights = None
if "--weights" in argv:
    i = argv.index("--weights")
    weights = json.load(open(argv[i + 1]))
    del argv[i:i + 2]
path = argv[0]
want = argv[1] if len(argv) > 1 else "whitted_kernelILi0ELb0ELb0ELb0ELb0E"

text = open(path).read()
m = re.search(r"^(_Z\w*" + re.escape(want) + r"\w*):\s*(;.*)?\n", text, re.M)
if not m:
    sys.exit(f"kernel {want} not found")
body = text[m.end(): text.index(".Lfunc_end", m.end())]

phase, before_fallback = "prologue", "prologue"
valu = collections.Counter()
trans = collections.Counter()
f64 = collections.Counter()
salu = collections.Counter()
mem = collections.Counter()
order = []
for line in body.split("\n"):
    t = line.strip()
    mm = re.match(r"; RT_PHASE (\w+)", t)
    if mm:
        name = mm.group(1)
        if name == "fallback":
            if phase != "fallback":
                before_fallback = phase
            phase = "fallback"
        elif name == "resume":
            if phase == "fallback":
                phase = before_fallback
        else:
            phase = name
        if phase not in order:
            order.append(phase)
        continue
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        valu[phase] += 1
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
            trans[phase] += 1
        if "f64" in op:
            f64[phase] += 1
    elif op.startswith(("s_load", "s_buffer", "global_", "flat_", "ds_", "buffer_", "scratch_")):
        mem[phase] += 1
    elif op.startswith("s_"):
        salu[phase] += 1

print(f"kernel {m.group(1)}")
hdr = f"{'phase':18s}{'VALU':>7s}{'f64':>7s}{'trans':>7s}{'SALU':>7s}{'mem':>6s}"
if weights:
    hdr += f"{'wave entries':>15s}{'VALU executed':>16s}{'share':>8s}"
print(hdr)
tot = collections.Counter()
rows = []
for ph in order:
    row = [ph, valu[ph], f64[ph], trans[ph], salu[ph], mem[ph]]
    if weights:
        w = weights["entries"].get(ph)
        row += [w, (w or 0) * valu[ph]]
    rows.append(row)
total_exec = sum(r[7] for r in rows) if weights else 0
for r in rows:
    line = f"{r[0]:18s}{r[1]:7d}{r[2]:7d}{r[3]:7d}{r[4]:7d}{r[5]:6d}"
    if weights:
        line += f"{(r[6] if r[6] is not None else 0):15.0f}{r[7]:16.0f}{(r[7] / total_exec if total_exec else 0):8.3f}"
    print(line)
print(f"{'total':18s}{sum(valu.values()):7d}{sum(f64.values()):7d}{sum(trans.values()):7d}{sum(salu.values()):7d}{sum(mem.values()):6d}", end="")
if weights:
    print(f"{'':15s}{total_exec:16.0f}   measured SQ_INSTS_VALU {weights.get('measured_valu', 0):.0f}")
else:
    print()
