python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_ring" 2>&1 | tail -3
for sh in 1,2 1,3 1,4 2,2; do echo "== shape $sh"; NSK_RING_SHAPE=$sh python scripts/time_ring.py 600,200 20 2>&1 | grep "ring  :\|walker"; done
echo "== trace 2,2"; NSK_RING_TRACE=1 NSK_RING_SHAPE=2,2 python scripts/time_ring.py 600,200 2 2>&1 | grep -A17 "ring trace" | head -36
echo "== trace 1,2"; NSK_RING_TRACE=1 NSK_RING_SHAPE=1,2 python scripts/time_ring.py 600,200 2 2>&1 | grep -A9 "ring trace" | head -20
