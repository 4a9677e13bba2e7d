for w in 8 9; do echo "--- config3 width $w"; SLIMM_GROUP_WIDTH=$w CONFIG=config3 bash scripts/group_libs.sh glibs8_$w "base"; done
for w in 8 10; do echo "--- config4 width $w"; SLIMM_GROUP_WIDTH=$w CONFIG=config4 bash scripts/group_libs.sh glibs8b_$w "base"; done
