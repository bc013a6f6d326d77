#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06v
mkdir -p $O
timeout -k 10 400 python tools/kappa_population.py --scene3d --kappa -1 1 0.5 0.25 > $O/kappa_population_scene3d.json 2> $O/kappa_population_scene3d.log; echo rc=$?; tail -3 $O/kappa_population_scene3d.log; cat $O/kappa_population_scene3d.json
