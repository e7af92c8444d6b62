# one gpurun call of round 5 (kept as the record of what produced gpurun_out/r05_b and the profiles/r05_* files derived from it)
out=gpurun_out/r05_b; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 && echo smoke ok || { echo smoke FAILED; tail -5 $out/smoke.log; }
(cd /tmp && /root/repo/dwarf_p_cloudsc2_tl_ad_amd/fortran/build/dwarf-cloudsc2-nl 1 160000 128 > /root/repo/$out/main_nl_default.out 2> /root/repo/$out/main_nl_default.err); tail -4 $out/main_nl_default.err
