"""Records the parity figures the GPU tests measure (agreement rates, error maxima) in
gpurun_out/parity_measured[_<engine>].json, so that the thresholds in the tests can be held at <= 2x what
was actually measured (the round's file is committed as profiles/roundN_parity_measured.json)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# a run under C3D_MATRIX=<engine> keeps its own report: the figures of the two engines differ
_ENGINE = os.environ.get("C3D_MATRIX")
PATH = os.path.join(ROOT, "gpurun_out", f"parity_measured_{_ENGINE}.json" if _ENGINE else "parity_measured.json")


def record(name, value):
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        data = json.load(open(PATH)) if os.path.exists(PATH) else {}
        data[name] = value
        json.dump(data, open(PATH, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    return value
