"""The sequence of sweep launches of tools/pmc_workload.py, shared with tools/pmc_parse.py: (label, repetitions).
Kernel names alone do not tell the launches apart (the adjoint's forward sweep without the evaporation branch IS the NL
kernel), so the parser walks the profiler's dispatches in order against this plan."""
REPS = 3
PLAN = [("satur", REPS), ("nl", REPS), ("tl", REPS), ("ad", REPS), ("ad_assign", REPS), ("ad_reverse", REPS),
        ("ad_reverse_assign", REPS)]
