for r in 1 2; do
bash scripts/group_libs.sh glibs7 "base gplain"
echo "--- staged"; SLIMM_GROUP_STAGED=1 bash scripts/group_libs.sh glibs7b "base gplain"
done
echo "--- config3"; CONFIG=config3 bash scripts/group_libs.sh glibs7d "base gplain"
echo "--- config3 staged"; SLIMM_GROUP_STAGED=1 CONFIG=config3 bash scripts/group_libs.sh glibs7e "base gplain"
