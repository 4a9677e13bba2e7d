python -m pytest tests/test_gpu_group_by_ident.py -m gpu -x -q 2>&1 | tail -3
bash scripts/group_libs.sh glibs4 "base"
echo "--- staged"; SLIMM_GROUP_STAGED=1 bash scripts/group_libs.sh glibs4b "base"
echo "--- width 9"; SLIMM_GROUP_WIDTH=9 bash scripts/group_libs.sh glibs4c "base"
echo "--- width 11"; SLIMM_GROUP_WIDTH=11 bash scripts/group_libs.sh glibs4f "base"
echo "--- config3"; CONFIG=config3 bash scripts/group_libs.sh glibs4d "base"
echo "--- config3 staged"; SLIMM_GROUP_STAGED=1 CONFIG=config3 bash scripts/group_libs.sh glibs4e "base"
