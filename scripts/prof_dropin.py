import cProfile, pstats, runpy, sys, io
sys.argv = ["scripts/time_dropin.py"]
pr = cProfile.Profile(); pr.enable()
runpy.run_path("scripts/time_dropin.py", run_name="__main__")
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
