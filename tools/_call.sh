bash tools/final_profile.sh 2>&1 | tail -8
