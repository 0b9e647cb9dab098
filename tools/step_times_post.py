"""print the `measured` block of a bench line on stdin, one field per line"""
import json, sys
r = json.loads(sys.stdin.readline())
print(json.dumps(r['measured'], indent=1)); print(r['value'], r['ms_per_step'], r['roofline'].get('frac_wall'))
