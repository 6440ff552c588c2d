"""measurement aid: which thread of a bench-like process burns CPU, and in which system call it sits.  Runs the resident step in a loop on a helper thread while the
main thread samples /proc/self/task/*/{stat,syscall,wchan}."""
import collections, os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
import bench

class A: pass
a = A(); a.contigs = 20; a.contig_len = 5_000_000; a.reads_per_contig = 2000; a.read_len = 15000; a.window = 750_000
mine = list(range(20))
contigs, blob, off, read_ctg = bench.make_inputs(2, mine, a.contig_len, lambda ci: 2000, a.read_len, a.window, 8)
ids = ["%06dF" % c for c in mine]
name_tab, maps = bench.make_names_and_maps(read_ctg, off, ids, 0)
from falcon_unzip_amd import _lib
eng = _lib.Engine(0)
job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
import tempfile, shutil
root = tempfile.mkdtemp(prefix="who_", dir="/dev/shm")
stop = False
def loop():
    k = 0
    while not stop:
        job.phase_write(ids, names=name_tab, out_dir=os.path.join(root, "s%d" % k), read_maps=maps, ctg_index=mine, async_writes=True, rebuild_index=True)
        k += 1
    eng.synchronize(); eng.pipe_flush()
th = threading.Thread(target=loop); th.start()
time.sleep(1.0)
tck = os.sysconf("SC_CLK_TCK")
def cpu():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % t).read()
            fld = st[st.rindex(")") + 2:].split()
            out[t] = (st[st.index("(") + 1:st.rindex(")")], int(fld[11]) / tck, int(fld[12]) / tck)
        except OSError:
            pass
    return out
c0 = cpu(); t0 = time.time()
samples = collections.defaultdict(collections.Counter)
for _ in range(3000):
    for t in os.listdir("/proc/self/task"):
        try:
            sc = open("/proc/self/task/%s/syscall" % t).read().split()
            wc = open("/proc/self/task/%s/wchan" % t).read().strip()
            samples[t][(sc[0] if sc else "?", sc[2] if len(sc) > 2 and sc[0] == "16" else "", wc)] += 1
        except OSError:
            pass
    time.sleep(0.001)
c1 = cpu(); dt = time.time() - t0
stop = True; th.join()
rows = sorted(((c1[t][1] - c0.get(t, (0, 0, 0))[1] + c1[t][2] - c0.get(t, (0, 0, 0))[2], t) for t in c1), reverse=True)
for d, t in rows[:6]:
    print("tid %s %s: %.1f %% of a core (user %.1f %%, system %.1f %%); top states (syscall nr, ioctl cmd, wchan):" % (t, c1[t][0], 100 * d / dt, 100 * (c1[t][1] - c0.get(t, (0, 0, 0))[1]) / dt, 100 * (c1[t][2] - c0.get(t, (0, 0, 0))[2]) / dt), samples[t].most_common(5))
shutil.rmtree(root, ignore_errors=True)
