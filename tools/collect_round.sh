#!/bin/bash
# After tools/gpu_round.sh TAG ... came back (gpurun merges gpurun_out/): copy the summaries the judge reads into profiles/.
# Usage: tools/collect_round.sh TAG
TAG=$1
for W in camera camera_zc camera_k20 camera_mesh lidar decay multicam multicam8 node; do
  S=""; [ $W != camera ] && S="_$W"
  [ -d gpurun_out/$TAG/stats$S ] || continue
  if [ -d gpurun_out/$TAG/pmc_fetch$S ]; then
    python tools/summarize_profile.py $TAG gpurun_out/$TAG/stats$S gpurun_out/$TAG/pmc_fetch$S gpurun_out/$TAG/pmc_write$S --workload $W > /dev/null
  else
    python tools/summarize_profile.py $TAG gpurun_out/$TAG/stats$S --workload $W
  fi
  [ -s gpurun_out/$TAG/bench$S.json ] && cp gpurun_out/$TAG/bench$S.json profiles/${TAG}_bench$S.json
done
ls profiles | grep $TAG
