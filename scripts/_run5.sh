for w in 4 6 7; do echo "--- width $w"; SLIMM_GROUP_WIDTH=$w bash scripts/group_libs.sh glibs5_$w "base"; done
for w in 4 6 ; do echo "--- staged width $w"; SLIMM_GROUP_STAGED=1 SLIMM_GROUP_WIDTH=$w bash scripts/group_libs.sh glibs5s_$w "base"; done
echo "--- grid 256"; SLIMM_GROUP_GRID=256 bash scripts/group_libs.sh glibs5g "base"
