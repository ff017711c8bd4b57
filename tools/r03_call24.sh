#!/bin/bash
# round 3, twenty-fourth GPU call: rocprofv3 kernel trace + PMC passes of the final library under the default workload
cd $GRAFT_REPO_ROOT
bash tools/profile_r03.sh final
