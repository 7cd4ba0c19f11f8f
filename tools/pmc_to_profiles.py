"""gpurun_out/pmc/mfma_summary.json (tools/pmc_mfma.sh) -> profiles/r01_pmc_all_kernels.json + r01_pmc_hbm_traffic.json"""
import json, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(os.path.join(ROOT, "gpurun_out/pmc/mfma_summary.json")))
json.dump(s, open(os.path.join(ROOT, "profiles/%s_pmc_all_kernels.json" % TAG), "w"), indent=1)
out = {"_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_mfma.sh, "
                "eager launches of the bench workload). Both counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of "
                "wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section), so read bytes = 2*FETCH_SIZE*1024 is an upper "
                "estimate for kernels that stream with 16-byte lanes and is used uniformly here; WRITE_SIZE is taken as reported. "
                "Inputs of ~100 MB stay resident in the 256 MiB Infinity Cache between launches, so small FETCH values are expected.",
       "kernels": {}}
for k, v in s.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out["kernels"][k] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                             "traffic_bytes": 2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024}
json.dump(out, open(os.path.join(ROOT, "profiles/%s_pmc_hbm_traffic.json" % TAG), "w"), indent=1)
print(len(out["kernels"]), "kernels")
