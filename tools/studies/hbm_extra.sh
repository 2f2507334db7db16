D=gpurun_out/r04q; mkdir -p $D; R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$D/pfe -o f -- python $R/bench.py --workload efficient_b256 --steps 2 > $R/$D/pfe.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$D/pwe -o w -- python $R/bench.py --workload efficient_b256 --steps 2 > $R/$D/pwe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$D/pfs -o f -- python $R/bench.py --workload stream128 > $R/$D/pfs.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$D/pws -o w -- python $R/bench.py --workload stream128 > $R/$D/pws.log 2>&1
cd $R
python profiles/summarize_pmc.py $(find $D/pfe -name "*.db") $(find $D/pwe -name "*.db") $D/efficient_hbm_traffic.json > $D/efficient_hbm.txt 2>&1
python profiles/summarize_pmc.py $(find $D/pfs -name "*.db") $(find $D/pws -name "*.db") $D/stream128_hbm_traffic.json > $D/stream128_hbm.txt 2>&1
head -14 $D/efficient_hbm.txt; head -14 $D/stream128_hbm.txt
find $D -name "*.db" -delete
