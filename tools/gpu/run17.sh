tools/prof_rank_static.sh 8 3 4096 2>&1 | tail -75
