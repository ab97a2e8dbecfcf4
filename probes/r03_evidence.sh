#!/bin/bash
# round 3 evidence pass 1: profiles of every benchable workload as the code stands now
set -o pipefail
R=$(pwd)
for WL in C2 C3 C4r; do bash probes/r03_profile.sh $WL || exit 1; done
TAG=C4opt_only1 bash probes/r03_profile.sh C4opt --opt-only 1 || exit 1
