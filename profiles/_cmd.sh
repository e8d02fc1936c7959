mkdir -p gpurun_out/s2u
for t in p0 p1; do
MFSGD_LIBRARY=profiles/_prof/libmfsgd_dev_$t.so bash profiles/ab.sh gpurun_out/s2u "${t}_nf:" "${t}_hc06:--hot-cost 0.6" "${t}_ml25m:--workload ml25m" >> gpurun_out/s2u/ab.txt 2>&1
done
