"""Assembles profiles/<tag>_*.md from the scratch output of tools/profile_round2.sh (gpurun_out/prof_<tag>/ and
gpurun_out/pmc_<tag>_*): python tools/make_profile_docs.py <tag> [atomic_pmc_dir]"""
import io, json, os, subprocess, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary

tag = sys.argv[1]
atomic_pmc = sys.argv[2] if len(sys.argv) > 2 else None
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()


def cap(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def read(name):
    p = os.path.join(src, name)
    return open(p).read() if os.path.exists(p) else f"(missing: {name})\n"


line = read("bench_line.json").strip().splitlines()[-1]
open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line + "\n")
d = json.loads(line)
md = [f"# {tag} — end of round 2: rocprofv3 evidence (MI355X, commit {head})\n",
      "All of it collected by `tools/profile_round2.sh` in one `gpurun` call; raw rocpd databases stay in `gpurun_out/` (scratch).\n",
      "## default bench line (`python bench.py`)\n```\n" + line + "\n```\n",
      f"Headline: {d['value']/1e9:.2f} G arcs/s, {d['ms_per_step']} ms per step (mean {d['ms_per_step_stats']['mean']} ± {d['ms_per_step_stats']['std']} ms over {d['steps']} steps); "
      f"`shortest_path(T)` alone {d['ms_shortest_path_T']} ms, fused batch alone {d['ms_compose_shortest_path_batch']} ms.\n",
      "## kernel table: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras` (both requests overlapped)\n",
      cap(rocpd_summary.trace, os.path.join(src, "trace_results.db")),
      "\n## the relaxation alone: `rocprofv3 --kernel-trace -- python tools/sp_repeat.py 1000000 12` (un-profiled solves, nothing else on the GPU)\n",
      cap(rocpd_summary.trace, os.path.join(src, "sp_alone_results.db")).split("\n\n")[-1] if True else "",
      "\n### timeline of one of those solves (launch order; duration and idle gap before each kernel)\n",
      cap(rocpd_summary.timeline, os.path.join(src, "sp_alone_results.db"), -3),
      "\n## where a sweep's time goes: phase stamps of `sssp_mbox_kernel` (`WFST_SSSP_MBOX_TRACE`, `tools/mbox_phases.py`)\n",
      "Per sweep: blocks that did something / slept, states expanded, messages sent, then for every phase boundary the time since the "
      "first block of the sweep started at which the LAST (median) busy block passed it: `trip1` = counts + keys + offsets + threshold "
      "have arrived, `applied` = inbox messages applied (LDS atomicMin), `scanned` = changed states written back and listed, `staged` = arc "
      "rows read and candidates staged, `expanded` = staged messages flushed, `end` = counts / waiting set published.  The kernel's "
      "duration in the timeline above adds ≈ 1.5 µs (tiny sweeps) to ≈ 3 µs (write-back of ≈ 10 MB of dirty lines) of launch / teardown.\n```\n",
      read("mbox_phases.txt"), "```\n",
      "\n## atomic sweeps vs mailbox sweeps, per-launch (profiled solves: one launch at a time, events around it)\n```\n", read("sweep_compare.txt"), "```\n"]
open(os.path.join(dst, f"{tag}_end_of_round.md"), "w").write("\n".join(md))

# counters
md = [f"# {tag} — PMC counters (rocprofv3 --pmc, one counter group per pass; commit {head})\n",
      "`tools/pmc_relax.sh`: every pass runs `tools/sp_repeat.py 1000000 6` (six un-profiled `shortest_path(T)` solves on the C3 graph); "
      "sums are PER SOLVE.  SQ_* counters are per shader engine instance summed over instances; SQ_WAVE_CYCLES / SQ_WAIT_* / "
      "SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).  FETCH_SIZE / WRITE_SIZE are KB.\n",
      "## `sssp_mbox_kernel` (mailbox sweeps, the default on this graph)\n```\n", read("pmc_mbox.txt"), "```\n"]
if atomic_pmc:
    md += ["## `sssp_relax_kernel` (atomic sweeps, `WFST_SSSP_MAILBOX=0`), same passes\n```\n",
           subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), atomic_pmc]).decode(), "```\n"]
md += ["## `string_compose_sp_kernel` (fused batch: 64 linear acceptors of 200 labels against the 1M-state T; `tools/pmc_batch.sh`)\n```\n",
       read("pmc_string.txt"), "```\n",
       "Reading (per wave and BFS level): the kernel is one dependent chain per problem — per level ≈ 62 VALU + 80 SALU + 2 VMEM + 4 LDS "
       "instructions of a lone wave (≈ 5 cycles each) and ONE dependent miss (TCC_MISS ≫ TCC_HIT: the arc block and its `anext` words come "
       "from beyond the L2 in one trip); SQ_WAIT_ANY against SQ_ACTIVE_INST_ANY gives the split (≈ 63 % / 36 % when it was measured with one "
       "wave per workgroup, `profiles/r02k`).  Since then a workgroup holds 8 problems (DESIGN.md §3.1): the SQ rows are per shader-engine "
       f"instance and now cover several waves each.  Kernel time per BFS level in the bench line of this set: {d.get('batch_kernel', {}).get('us_per_bfs_level')} µs.\n"]
open(os.path.join(dst, f"{tag}_counters.md"), "w").write("\n".join(md))

# wide compose
md = [f"# {tag} — the wide composition driver (commit {head})\n",
      "`rocprofv3 --kernel-trace -- python tools/lookahead_timing.py 10000,100,3,16,16`: the look-ahead composition (1.22 M states / "
      "4.27 M arcs), the plain composition without connect (1.11 M states / 3.9 M arcs) and the default `compose()` (with connect) of the "
      "same pair, each once as warm-up and three times timed.  The driver as of this set (DESIGN.md §3.7): 8 lanes per composed state, "
      "three launches per level (`la_emit`, `la_first`, `la_assign`) that read the level's id range from the control block, up to 8 "
      "levels queued per host look, the arena grown in place and ahead of the level that would not fit.\n",
      cap(rocpd_summary.trace, os.path.join(src, "wide_results.db")),
      "\n## end-to-end times of the tool (warm), oracle-identical where the oracle was run\n```\n", read("wide_timing.txt"), "```\n",
      "(gpu_ms = look-ahead composition on the GPU, cpu_ms = the oracle's, plain_gpu_ms = `compose(connect=False)` of the same pair, the last "
      "column the default `compose()` with connect; states in parentheses; best of three.)  Round 1: 4.2 / 5.4 / 8.5 / 27.8 ms for the four "
      "plain compositions, 1.7 / 4.4 / 11.4 / 39.7 ms with look-ahead.\n"]
open(os.path.join(dst, f"{tag}_wide_compose.md"), "w").write("\n".join(md))
for name in ("kdelta_gap.txt", "rm_epsilon_timing.txt", "step_breakdown.txt"):  # (optional parts of the set)
    if os.path.exists(os.path.join(src, name)):
        open(os.path.join(dst, f"{tag}_{name}"), "w").write(read(name))
print("written")
