"""Collects the parity error every GPU test measured (test -> worst rel-L2, its limit) so the margins are visible:
written at session end to gpurun_out/parity_r06.json (copied into profiles/ for the record).  EDM_PARITY_LOG=0 (child
pytest processes that re-run a subset under another kernel selection) keeps a session from writing."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_records = {}


def record(name: str, err: float, limit: float) -> None:
    prev = _records.get(name)
    if prev is None or err > prev["err"]:
        # margin = limit / err; an exact-equality check (limit 0) that measured 0 has no finite margin: recorded as null
        margin = None if (limit == 0 and err == 0) else float(limit) / max(float(err), 1e-30)
        _records[name] = {"err": float(err), "limit": float(limit), "margin": margin}


def dump() -> None:
    if not _records or os.environ.get("EDM_PARITY_LOG") == "0":
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "parity_r06.json")
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(_records)
        with open(path, "w") as f:
            json.dump(dict(sorted(old.items())), f, indent=1)
    except OSError:
        pass
