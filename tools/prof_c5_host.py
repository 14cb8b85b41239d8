import os, sys, cProfile, pstats, io
sys.argv=["x"]
ROOT="/root/repo"
sys.path.insert(0, ROOT)
src=open(os.path.join(ROOT,"tools","time_c5_phases.py")).read()
# strip the timing wrappers and the loops: reuse the setup
setup=src.split("acc = {}")[0]
exec(setup)
for _ in range(3):
    pipeline.quantify(dcool, positions, cfg, inter=True, max_dist_bp=md)
pr=cProfile.Profile()
pr.enable()
for _ in range(20):
    pipeline.quantify(dcool, positions, cfg, inter=True, max_dist_bp=md)
pr.disable()
s=io.StringIO()
pstats.Stats(pr,stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
