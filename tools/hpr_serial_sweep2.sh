for ids in 192 256 320 384 448; do echo "== $ids"; TOHIP_HULL_SERIAL_IDS=$ids python3 tools/hpr_once.py 1000000 12 2>/dev/null | tail -1; done
for ids in 640 768 896; do echo "== batched $ids"; TOHIP_HULL_SERIAL_IDS=$ids python3 tools/hpr_batched_once.py 4 2>/dev/null | tail -1; done
