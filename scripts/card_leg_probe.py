"""Dev: bench.py's .card / raw -> .toad leg on its own, twice in one process."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

for rep in range(2):
    out = bench.card_to_toad_leg(int(sys.argv[1]) if len(sys.argv) > 1 else 65536)
    print(json.dumps({k: out[k] for k in ("gpu_blocks_per_s", "gpu_loop_blocks_per_s", "raw_gpu_blocks_per_s",
                                          "gpu_construct_s", "gpu_loop_stats", "raw_gpu_loop_stats")}))

