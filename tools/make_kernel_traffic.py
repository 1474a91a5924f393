"""profiles/<tag>_summary.json (tools/summarize_prof.py, PMC passes) -> profiles/kernel_traffic.json, the per-launch HBM
bytes bench.py quotes as roofline.traffic.  usage: python tools/make_kernel_traffic.py r01_final"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
s = json.load(open(f"profiles/{tag}_summary.json"))
f, w = s["pmc_raw_KB_per_launch"]["FETCH_SIZE"], s["pmc_raw_KB_per_launch"]["WRITE_SIZE"]
want = {"fpv::rollout_step_kernel": "rollout_step_kernel",
        "fpv::rollout_step_split_kernel": "rollout_step_split_kernel",
        "fpv::policy_split_forward_kernel": "policy_split_forward_kernel",
        "fpv::policy_forward_kernel": "policy_forward_kernel<512>",
        "fpv::step_quad_kernel": "step_quad_kernel",
        "fpv::step_kernel<float,true,true>": "step_kernel<float, true, true",
        "fpv::step_kernel<float,true,false>": "step_kernel<float, true, false",
        "fpv::gae_kernel": "fpv::gae_kernel",
        "fpv::ledger_scan_kernel": "ledger_scan_kernel",
        "fpv::tile_partials_merge_kernel": "tile_partials_merge_kernel"}
out = {"drones": 65536, "round": tag, "kernels": {}, "source": f"profiles/{tag}_summary.json",
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py; FETCH_SIZE doubled per "
                 "MI355X_MICROARCH.md §HBM (gfx950 reports half of a wide coalesced read); KB -> bytes; mean over the "
                 "launches of the pass (the policy kernel's mean includes the rollout tail's two short launches per 32 steps)"}
for key, needle in want.items():
    src = [k for k in f if needle in k]
    if not src:
        continue
    k = src[0]
    rd, wr = int(2 * f[k]["mean_KB"] * 1024), int(w.get(k, {"mean_KB": 0})["mean_KB"] * 1024)
    out["kernels"][key] = {"read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                           "launches_in_pass": f[k]["launches"], "symbol": k}
json.dump(out, open("profiles/kernel_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
