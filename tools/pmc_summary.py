"""Per-kernel, per-solve sums of every counter found in the rocpd databases of a tools/pmc_relax.sh run."""
import glob, os, sqlite3, sys
out = sys.argv[1]
rows = {}
for db in sorted(glob.glob(os.path.join(out, "*_results.db"))):
    c = sqlite3.connect(db)
    try:
        n_solves = c.execute("select count(*) from pmc_events where name like '%setup_kernel%' group by counter_name").fetchone()
    except sqlite3.Error as e:
        print(f"{db}: {e}")
        continue
    n_solves = max(1, n_solves[0] if n_solves else 1)
    for name, cname, n, tot in c.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
        short = name.replace("wfst::(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0]
        if os.environ.get("PMC_KERNELS") == "all":
            if not short.startswith("sssp_") or "setup" in short:
                continue
        elif short not in ("sssp_relax_kernel", "sssp_mbox_kernel", "sssp_mbox_resident_kernel"):
            continue
        rows.setdefault(short, {})[cname] = (n / n_solves, tot / n_solves)
for k, cs in rows.items():
    print(f"## {k} (per solve)")
    for cname, (n, tot) in sorted(cs.items()):
        print(f"  {cname:28s} launches {n:6.1f}  sum {tot:16.1f}")
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f, w = cs["FETCH_SIZE"][1] * 1024, cs["WRITE_SIZE"][1] * 1024  # KB -> bytes
        print(f"  -> traffic per solve: FETCH raw {f/1e6:.1f} MB (x2 per MI355X_MICROARCH.md: {2*f/1e6:.1f} MB), WRITE {w/1e6:.1f} MB; "
              f"2xFETCH+WRITE = {(2*f+w)/1e6:.1f} MB = {(2*f+w)/212e6:.2f} x algorithmic (212 MB); raw FETCH+WRITE = {(f+w)/212e6:.2f} x")
    if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
        h, m = cs["TCC_HIT_sum"][1], cs["TCC_MISS_sum"][1]
        print(f"  -> L2 hit rate {h/(h+m):.3f}")
