"""The CIN rows of tools/candidates.py alone (hk_cin_* at configs/CIN.yaml's shape).      python tools/cin_rows.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import candidates as c

c.guarded(c.cin)
print(json.dumps(c.rows, indent=1), flush=True)
