"""Reduce the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into
profiles/pmc_traffic.json: HBM bytes per launch of the three dominant conv kernel families.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py ...
    python scripts/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are
in KiB; FETCH_SIZE tallies the 128-B requests of 16-B/lane streaming reads at 64 B, so it is doubled.
"""
import csv, glob, json, os, sys
from collections import defaultdict

# round 3: the V-form kernels (tonal_wino43v.hip); names as Engine.kernel_families() / rocprofv3 give them.  A weight-
# gradient op is two launches (the Vd-writing first C_in tile, then the other tiles), listed apart.
F6 = "Winograd F(6,3) on pre-transformed operands, LDS-DMA"
FAMILIES = {
    # round 4: the F(6,3) kernels (tonal_wino63.hip), the default
    "wino63v_nt_kernel<5>": f"wino63v_nt_kernel<POOLV> (conv2 forward, {F6}; writes V of its pooled output for conv3)",
    "wino63v_nt_kernel<2>": f"wino63v_nt_kernel<POOL> (conv3 forward, {F6})",
    "wino63v_nt_kernel<4>": f"wino63v_nt_kernel<C1WGRAD> (conv2 input gradient + conv1 weight gradient, {F6})",
    "wino63v_nt_kernel<6>": f"wino63v_nt_kernel<MASKY> (conv3 input gradient, {F6}; writes Y and Vd of conv2 instead of the gradient rows)",
    "wino63v_nt_kernel<3>": f"wino63v_nt_kernel<MASK> (conv3 input gradient, {F6})",
    "wino63v_nt_kernel<7>": "wino63v_nt_kernel<GY> (conv4 input gradient as six batched GEMMs of the NT63 kernel; writes Y and Vd of conv3 instead of the gradient rows)",
    "wino63_unpool_rows6_kernel": "wino63_unpool_rows6_kernel (conv4's pooled output gradient un-pooled into the hex-slot operand of its input gradient)",
    "wino63v_tn4y_kernel": f"wino63v_tn4y_kernel (conv2 / conv3 weight gradient, {F6}: both operands by LDS-DMA, no transform in the kernel; average of the two launches)",
    "wino63_unpool_yvd_kernel": "wino63_unpool_yvd_kernel (Y3 / Vd3 of conv3 from G3 and its arg-max bits)",
    "conv1_fwd_vh_kernel": "conv1_fwd_vh_kernel (conv1 + LeakyReLU + pool writing V1 in hex form)",
    "nadam_lowrank_kernel<true>": "nadam_lowrank_kernel<true> (NAdam on W_hh from its gradient factors + the last BPTT product dh_1 in the same pass)",
    "tn_skinny_kernel": "tn_skinny_kernel (dgates . W_hh on the distinct label rows: one 5.4 GB pass)",
    "nt_window_kernel<32, 0, 0>": "nt_window_kernel<32, 0, 0> (h . W_hh^T on the distinct label rows: one 5.4 GB pass)",
    "wino63v_tn4_kernel<true>": f"wino63v_tn4_kernel<true> (conv3 weight gradient, {F6}; also writes Vd)",
    "wino43v_nt_kernel<5>": "wino43v_nt_kernel<POOLV> (conv2 forward, Winograd F(4,3) on V, LDS-DMA; writes V of its pooled output for conv3 instead of the raw rows)",
    "wino43v_nt_kernel<2>": "wino43v_nt_kernel<POOL> (conv3 forward, Winograd F(4,3) on the pre-transformed input V, LDS-DMA)",
    "wino43v_nt_kernel<4>": "wino43v_nt_kernel<UNPOOL,C1WGRAD> (conv2 input gradient + conv1 weight gradient, Winograd F(4,3) on the pre-transformed dZ, LDS-DMA)",
    "wino43v_nt_kernel<3>": "wino43v_nt_kernel<UNPOOL,MASK> (conv3 input gradient, Winograd F(4,3) on the pre-transformed dZ, LDS-DMA)",
    "wino43v_tn8_kernel<true>": "wino43v_tn8_kernel<true> (conv2/conv3 weight gradient, Winograd F(4,3) on V, LDS-DMA; also writes Vd for the input gradient)",
    "wino43v_tn_kernel<false": "wino43v_tn_kernel<false, .> (conv2/conv3 weight gradient, the other C_in tiles, Winograd F(4,3) on V, LDS-DMA)",
    "wino43v_tn_kernel<true": "wino43v_tn_kernel<true, .> (conv2/conv3 weight gradient, C_in tile 0, + writes Vd for the input gradient)",
    "wino43_nt_kernel<1, 4": "wino43_nt_kernel<UNPOOL,C1WGRAD> (conv2 input gradient + conv1 weight gradient, Winograd F(4,3))",
    "wino43_nt_kernel<1, 3": "wino43_nt_kernel<UNPOOL,MASK> (conv3 input gradient, Winograd F(4,3))",
    "wino43_nt_kernel<0, 2": "wino43_nt_kernel<DIRECT,POOL> (conv2/conv3 forward, Winograd F(4,3))",
    "wino_nt_kernel<1, 3, 1>": "wino_nt_kernel<UNPOOL,MASK> (conv2/conv3 input gradient, Winograd F(2,3))",
    "wino_nt_kernel<0, 2, 1>": "wino_nt_kernel<DIRECT,POOL> (conv2/conv3 forward, Winograd F(2,3))",
    "wino_tn_kernel": "wino_tn_kernel (conv2/conv3 weight gradient, Winograd F(2,3))",
    "wino43_tn_kernel": "wino43_tn_kernel (conv2/conv3 weight gradient, Winograd F(4,3))",
}


# the two launches of wino63v_tn4y_kernel per step (conv3's, then conv2's: same kernel, same grid) are told apart by duration and
# ALSO reported under the names CnnEngine.kernel_families() gives them, so that bench.py finds its dominant kernel here
TN4Y = {"long": f"wino63v_tn4y_kernel (conv2 weight gradient, {F6}: both operands by LDS-DMA, no transform in the kernel)",
        "short": f"wino63v_tn4y_kernel (conv3 weight gradient, {F6}: both operands by LDS-DMA, no transform in the kernel; Y3 / Vd3 "
                 "from the epilogue of conv4's input gradient)"}


def tn4y_split(rows):
    """{Dispatch_Id: 'long' | 'short'} for the wino63v_tn4y_kernel dispatches of a counter_collection table"""
    dts = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows if "wino63v_tn4y_kernel" in r["Kernel_Name"]}
    if not dts:
        return {}
    mid = (max(dts.values()) + min(dts.values())) / 2
    return {k: ("long" if v > mid else "short") for k, v in dts.items()}


def collect(d, counter):
    tot, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        which = tn4y_split(rows)
        for r in rows:
            if r["Counter_Name"] != counter:
                continue
            for key, fam in FAMILIES.items():
                if key in r["Kernel_Name"]:
                    tot[fam] += float(r["Counter_Value"])
                    n[fam].add(r["Dispatch_Id"])
            if r["Dispatch_Id"] in which:
                fam = TN4Y[which[r["Dispatch_Id"]]]
                tot[fam] += float(r["Counter_Value"])
                n[fam].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in list(dict.fromkeys(FAMILIES.values())) + list(TN4Y.values()):
    if fam not in fetch or fam not in write:
        continue
    rd = fetch[fam] / nf[fam] * 1024 * 2
    wr = write[fam] / nw[fam] * 1024
    out[fam] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "launches_averaged": nf[fam],
                "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 tallies 128-B requests at 64 B for 16-B/lane streaming "
                              "reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE KiB x1024"}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
