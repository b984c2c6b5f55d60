"""profiles/pmc_relax_traffic.json from the rocpd databases of a tools/pmc_relax.sh run (read by bench.py for
roofline.traffic): python tools/pmc_to_json.py gpurun_out/pmc_<tag> <tag>   (run in the repository: records HEAD and the
hash of the relaxation sources, so that bench.py can tell a stale figure)."""
import glob, json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha256

out_dir, tag = sys.argv[1], sys.argv[2]
KERNELS = ("sssp_relax_kernel", "sssp_mbox_kernel", "sssp_mbox_resident_kernel")
per = {}
for db in sorted(glob.glob(os.path.join(out_dir, "*_results.db"))):
    c = sqlite3.connect(db)
    n_solves = c.execute("select count(*) from pmc_events where name like '%setup_kernel%' group by counter_name").fetchone()
    n_solves = max(1, n_solves[0] if n_solves else 1)
    for name, cname, n, tot in c.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
        short = name.replace("wfst::(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0]
        if short in KERNELS:
            per.setdefault(short, {})[cname] = (n / n_solves, tot / n_solves)
kernel = max(per, key=lambda k: per[k].get("FETCH_SIZE", (0, 0))[1])
# a solve's relaxation is every launch of these kernels (resident launches: the head and the tail are sssp_mbox_kernel launches):
# counters are summed over them, `kernel` names the one that moves most
cs = {}
for k, d in per.items():
    for cname, (n, tot) in d.items():
        a = cs.get(cname, (0.0, 0.0))
        cs[cname] = (a[0] + n, a[1] + tot)
per_kernel = {k: {c: round(v[1]) for c, v in d.items()} for k, d in per.items()}
fetch, write = cs["FETCH_SIZE"][1] * 1024, cs["WRITE_SIZE"][1] * 1024
res = {
    "kernel": kernel,
    "source": f"rocprofv3 --pmc passes of tools/profile_round.sh {tag} (or tools/pmc_relax.sh; each pass runs tools/sp_repeat.py: un-profiled shortest_path(T) solves), one counter group per pass",
    "workload": "T 1M states / 10M arcs, fan-out 10, seed 3 (bench.py default)",
    "commit": subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip(),
    "kernel_sources_sha256": kernel_sources_sha256(),
    "launches_per_solve": round(cs["FETCH_SIZE"][0], 1),
    "per_kernel_per_solve": per_kernel,
    "fetch_bytes_per_solve_raw": round(fetch),
    "write_bytes_per_solve": round(write),
    "traffic_bytes_per_solve": round(2 * fetch + write),
    "traffic_bytes_per_solve_uncorrected": round(fetch + write),
    "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B; calibrated there on wide coalesced reads, which is what this kernel issues except for the arc rows of expanded states)",
    "counters_file": f"profiles/{tag}_counters.md",
}
if "TCC_HIT_sum" in cs:
    res["l2_hit_rate"] = round(cs["TCC_HIT_sum"][1] / (cs["TCC_HIT_sum"][1] + cs["TCC_MISS_sum"][1]), 3)
if "TCC_EA0_ATOMIC_sum" in cs:
    res["tcc_ea_atomic_per_solve"] = round(cs["TCC_EA0_ATOMIC_sum"][1])
for k in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_LDS",
          "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE",
          "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_ATOMIC_sum"):
    if k in cs:
        res.setdefault("counters_per_solve", {})[k] = round(cs[k][1])
json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_relax_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
