# Debugging aid: repeats the device optimizer tests to expose run-to-run races (bash tools/flake_loop.sh on the GPU box)
export TMPDIR=/tmp
fails=0
for i in $(seq 1 ${LOOPS:-36}); do
  out=$(timeout 120 python -m pytest tests/test_gpu_optimizer.py -m gpu -q -x 2>&1)
  if echo "$out" | grep -q FAILED; then fails=$((fails+1)); echo "$out" | grep -E "^E |FAILED" | cut -c1-600 | head -12; fi
done
echo "$fails / ${LOOPS:-36} runs failed"
