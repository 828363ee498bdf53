"""The drop-in leg of bench.py alone (frame loop + planted loop), for a kernel trace of what a frame's match + filter costs."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
print(json.dumps(bench.dropin_leg(int(sys.argv[1]) if len(sys.argv) > 1 else 20))[:900])
