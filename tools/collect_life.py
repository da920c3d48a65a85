"""Merge the per-configuration lifetime summaries of tools/block_life.py --json (gpurun_out/r6/block_life_<cfg>.json, or $HSR_ROUND_DIR) into profiles/block_life.json.
usage: python tools/collect_life.py <note>"""
import json, os, sys
from pathlib import Path
root = Path(__file__).resolve().parents[1]
out = {}
for cfg in ("cfg1", "cfg2", "cfg3", "cfg4", "cupboard"):
    f = Path(os.environ.get("HSR_ROUND_DIR", str(root / "gpurun_out" / "r6"))) / f"block_life_{cfg}.json"
    if f.exists():
        out.update(json.loads(f.read_text()))
out["_note"] = sys.argv[1] if len(sys.argv) > 1 else ""
(root / "profiles" / "block_life.json").write_text(json.dumps(out, indent=1))
print({k: (round(v["p50_ms"], 2), round(v["p100_ms"], 2), round(v["mean_over_p100"], 3)) for k, v in out.items() if isinstance(v, dict)})
