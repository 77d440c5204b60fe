#!/usr/bin/env python3
"""Second half of tools/kernel_coverage.sh: kernel names dispatched in any rocprofv3 database under OUT/prof (one per traced process)
against OUT/all_kernels.txt (unit, demangled kernel name) -> OUT/launched.txt, OUT/never_launched.txt."""
import glob, os, re, sqlite3, subprocess, sys
out = sys.argv[1]
names, procs = {}, 0
for dbf in glob.glob(os.path.join(out, "prof", "**", "*.db"), recursive=True):
    db = sqlite3.connect(dbf)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    if not kd or not ks:
        continue
    procs += 1
    for n, c in db.execute("select s.kernel_name, count(*) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name" % (kd[0], ks[0])):
        names[n] = names.get(n, 0) + c
mangled = [re.sub(r"\.kd$", "", n) for n in names]
dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.split("\n")
launched = {}
for n, d in zip(names, dem):
    d = re.sub(r"\([^()]*\)$", "", re.sub(r"^void ", "", d))
    launched[d] = launched.get(d, 0) + names[n]
allk = [l.rstrip("\n").split(" ", 1) for l in open(os.path.join(out, "all_kernels.txt")) if " " in l]
with open(os.path.join(out, "launched.txt"), "w") as f:
    for u, k in allk:
        if k in launched:
            f.write("%-14s %8d  %s\n" % (u, launched[k], k))
never = [(u, k) for u, k in allk if k not in launched]
with open(os.path.join(out, "never_launched.txt"), "w") as f:
    for u, k in never:
        f.write("%-14s %s\n" % (u, k))
print("traced processes %d, kernels in the library %d, launched by the suite %d, never launched %d" % (procs, len(allk), len(allk) - len(never), len(never)))
for u, k in never:
    print("  never:", u, k)
