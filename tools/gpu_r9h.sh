#!/bin/bash
# Round 6, final validation session: full GPU suite, smoke, default-flag bench, rocprofv3 kernel statistics live and serial
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/gpu_session.sh r9h tests smoke bench kt serial
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r9h/bench.json')))
print(d['ms_per_step'], d['value'], json.dumps(d.get('synthesis_summary')))
PY
