mkdir -p gpurun_out/s3q
for t in pre1 base1; do
MFSGD_LIBRARY=profiles/_prof/libmfsgd_dev_$t.so bash profiles/ab.sh gpurun_out/s3q "${t}_ml1m:--workload ml1m" "${t}_ml1m_hm16:--workload ml1m --hot-min 16" "${t}_ml1m_hm24:--workload ml1m --hot-min 24"  "${t}_ml1m_hm8:--workload ml1m --hot-min 8" "${t}_nf64:--k 64" >> gpurun_out/s3q/ab.txt 2>&1
done
