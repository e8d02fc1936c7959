mkdir -p gpurun_out/s2y
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/s2y/tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/s2y/tests.log
tail -3 gpurun_out/s2y/tests.log
grep -q "rc=0" gpurun_out/s2y/tests.log || exit 1
bash profiles/ab.sh gpurun_out/s2y "nf:" "nf_frozen:--implicit frozen" "ml25m:--workload ml25m" "ml1m:--workload ml1m" "nf64:--k 64" "nf256:--k 256" "yahoo:--workload yahoo --steps 4" "ml100k:--workload ml100k" "nf_k40:--k 40" "nf_g320:--grid 320,254" "nf_g352:--grid 352,254" "ml25m_g320:--workload ml25m --grid 320,256" "ml25m_g272:--workload ml25m --grid 272,256" "ml1m_93:--workload ml1m --grid 93,75" "ml1m_110:--workload ml1m --grid 110,90" "yahoo_320:--workload yahoo --steps 4 --grid 320,5252" "nf_hc01:--hot-cost 0.1" "nf_hc05:--hot-cost 0.5" > gpurun_out/s2y/ab.txt 2>&1 || exit 1
for w in 2 8; do timeout -k 10 300 python profiles/dsgd_rank_sim.py $w >> gpurun_out/s2y/rank_sim.txt 2>&1 || exit 1; done
