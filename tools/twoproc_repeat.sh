# Repeats the two-process string-pass run of tests/test_gpu_two_process.py (two processes, each 200 x {stage 1, string pass} over 80 MB)
# and prints every mismatch: the liveness soak that found the scanner-by-workgroup-number hole of k_strings.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 32); do
  python tools/two_proc_worker.py procA 200 128 strings > /tmp/a.log 2>&1 &
  pa=$!
  python tools/two_proc_worker.py procB 200 128 strings > /tmp/b.log 2>&1 &
  pb=$!
  wait $pa; wait $pb
  grep -h -E "MISMATCH|bad" /tmp/a.log /tmp/b.log | grep -v " 0 bad" 
done
echo done
