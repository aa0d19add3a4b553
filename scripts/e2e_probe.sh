# End-to-end run of tests/cpp/gemma_file_driver on synthetic PLINK files (n = 20000, 200000 SNPs), alone and beside an idle
# process that holds a device context and memory (as bench.py's parent does during its e2e leg); eigensolver stage timings on.
set -e
cd $GRAFT_REPO_ROOT
T=/tmp/e2ep; rm -rf $T; mkdir -p $T
g++ -std=c++11 -O2 -Iinclude tests/cpp/io_host_check.cpp -lz -pthread -o $T/gen
g++ -std=c++11 -O2 -Iinclude tests/cpp/gemma_file_driver.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o $T/drv
$T/gen plinkgen $T/S 20000 200000 64 > /dev/null
run() { echo "== $1"; env $2 GEMMA_HIP_EIGH_TIMING=1 $T/drv -bfile $T/S -inproc 1 -lmm 1 -outdir $T -o e2e 2>&1 | grep -E "gemma_hip_eigh n=.*two-stage|t_first" | sed -e 's/trace_G.*t_null/t_null/' | cut -c1-330; }
run "alone" X=1
python -c "
import torch, time
x = torch.zeros(8 * 10**9 // 8, dtype=torch.float64, device='cuda'); torch.cuda.synchronize(); print('holder up', flush=True); time.sleep(45)" &
HP=$!
sleep 12
run "beside an idle context holder" X=1
run "beside an idle context holder, plain launch of the chase" GEMMA_HIP_EIGH_BC_COOP=0
kill $HP 2>/dev/null || true
wait $HP 2>/dev/null || true
