"""cProfile of the host side of a bench line: where the Python time of a step goes (top functions by own time).
usage: python tools/host_profile.py <bench args...>"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "40", "--warmup", "5"] + sys.argv[1:]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:9000], file=sys.stderr)
