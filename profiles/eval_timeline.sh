# Kernel timeline of one replayed batch of the captured mini-batch evaluation (rocprofv3 --kernel-trace of profiles/bench_eval.py --minibatch): bash profiles/eval_timeline.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/evtl && O=gpurun_out/evtl
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python profiles/bench_eval.py --minibatch --batches 300 > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 2; }
python - <<PY > $O/timeline_eval.txt
import glob, os, sqlite3, statistics
from collections import Counter
db = sorted(glob.glob(os.path.join("$O/prof", "**", "*_results.db"), recursive=True))[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
names=[r[0] for r in rows]
# a batch starts at step_begin / or find period: use the last 40% of the trace and the most common kernel-name period
tail = rows
first = Counter(r[0] for r in tail).most_common()
# period detection: positions of the rarest-but-repeated kernel
cand = [n for n,c in first if c >= 250]
key = min(cand, key=lambda n: Counter(r[0] for r in tail)[n])
idx = [i for i,r in enumerate(rows) if r[0]==key]
per = Counter(b-a for a,b in zip(idx, idx[1:])).most_common(1)[0][0]
st = [a for a,b in zip(idx, idx[1:]) if b-a==per]; st = st[len(st)//2:]
print("# key", key[:50], "period", per, "steps", len(st))
tot=0
for p in range(per):
    d = statistics.median(rows[a+p][2]-rows[a+p][1] for a in st)/1e3
    g = statistics.median(rows[a+p][1]-rows[a+p-1][2] for a in st)/1e3
    tot += d + max(g,0)
    print(f"{p:3d} {rows[st[0]+p][0][:70]:70s} dur {d:7.2f} gap {g:6.2f} t {tot:8.2f}")
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
cat $O/timeline_eval.txt | cut -c1-130
