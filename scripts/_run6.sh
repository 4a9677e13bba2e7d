bash scripts/group_libs.sh glibs6 "base"
echo "--- staged"; SLIMM_GROUP_STAGED=1 bash scripts/group_libs.sh glibs6b "base"
echo "--- config3"; CONFIG=config3 bash scripts/group_libs.sh glibs6d "base"
echo "--- config3 staged"; SLIMM_GROUP_STAGED=1 CONFIG=config3 bash scripts/group_libs.sh glibs6e "base"
