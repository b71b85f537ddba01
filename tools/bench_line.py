import json
import sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(tag, d["value"], d["ms_per_step"], d["stages_ms"])
