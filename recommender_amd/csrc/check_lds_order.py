#!/usr/bin/env python3
"""Build-time check of the LDS hand-offs in epoch.hip (MFSGD_LDS_FENCE = 0 build): python check_lds_order.py epoch.s

Every LDS_RELEASE() leaves the marker `; mf_lds_release` in the ISA and every LDS_ACQUIRE() `; mf_lds_acquire`.
Checked per kernel:
  * release: the payload was stored BEFORE the marker - the last DS instruction in front of it (same basic block
    permitting) is a ds_write - and the first DS instruction behind it is the ticket's ds_write (a plain store: relaxed
    workgroup-scope atomics on LDS are ds_write_b32);
  * acquire: the last DS instruction in front of the marker is the ticket's ds_read, the payload's reads follow it.
Fails (exit 1) on any other order, or if a kernel that stages its item stripe in LDS has no marker at all.
"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
ds = re.compile(r"^\s+(ds_\w+)")
bad, kernels, markers = [], 0, 0
name = None
body = []


def check(name, body):
    global markers
    n_rel = n_acq = 0
    for j, line in enumerate(body):
        if "mf_lds_release" in line:
            n_rel += 1
            nxt = next((ds.match(l).group(1) for l in body[j + 1:] if ds.match(l)), None)
            if nxt is None or not nxt.startswith("ds_write"):
                bad.append(f"{name}: after a release marker the first DS instruction is {nxt}, not the ticket's ds_write")
            prv = next((ds.match(l).group(1) for l in reversed(body[:j]) if ds.match(l)), None)
            if prv is not None and not (prv.startswith("ds_write") or prv.startswith("ds_read") or prv.startswith("ds_bpermute")
                                        or prv.startswith("ds_add") or prv.startswith("ds_max")):
                bad.append(f"{name}: unexpected DS instruction {prv} in front of a release marker")
        if "mf_lds_acquire" in line:
            n_acq += 1
            prv = next((ds.match(l).group(1) for l in reversed(body[:j]) if ds.match(l)), None)
            if prv is None or not prv.startswith("ds_read"):
                bad.append(f"{name}: in front of an acquire marker the last DS instruction is {prv}, not the ticket's ds_read")
    markers += n_rel + n_acq
    return n_rel, n_acq


for line in text:
    m = re.match(r"^(_Z\w*k_epoch_(free|fixed)\w*):", line)
    if m:
        if name:
            check(name, body)
        name, body = m.group(1), []
        kernels += 1
    elif line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        if name:
            rel, acq = check(name, body)
            # <T, KC, STAGE = true, ...>: staged kernels hand rows over through LDS
            if "Lb1E" in name and "k_epoch_free" in name and (rel == 0 or acq == 0):
                bad.append(f"{name}: a staged kernel without hand-off markers (release {rel}, acquire {acq})")
        name, body = None, []
    elif name:
        body.append(line)
if kernels == 0 or markers == 0:
    bad.append(f"found {kernels} epoch kernels and {markers} markers: is this the ISA of epoch.hip built with MFSGD_LDS_FENCE=0?")
for b in bad:
    print("check_lds_order:", b)
print(f"check_lds_order: {kernels} epoch kernels, {markers} hand-off markers, {len(bad)} violations")
sys.exit(1 if bad else 0)
