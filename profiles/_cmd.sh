mkdir -p gpurun_out/s3o
MFSGD_LIBRARY=profiles/_prof/libmfsgd_dev_pre.so timeout -k 10 300 python profiles/dev_walker.py > gpurun_out/s3o/dev_walker.txt 2>&1 || { tail -5 gpurun_out/s3o/dev_walker.txt; exit 1; }
MFSGD_LIBRARY=profiles/_prof/libmfsgd_dev_pre.so bash profiles/ab.sh gpurun_out/s3o "pre_nf:" "pre_ml25m:--workload ml25m" "pre_frozen:--implicit frozen" "pre_ml25m_hm32:--workload ml25m --hot-min 32" >> gpurun_out/s3o/ab.txt 2>&1
