"""bench.py's `dropin` leg on its own (frame_loop / value / slam_loop through the reference's names): one JSON object."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
print(json.dumps(bench.dropin_leg(int(sys.argv[1]) if len(sys.argv) > 1 else 96), indent=1))
