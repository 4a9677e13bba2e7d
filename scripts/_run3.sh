python -m pytest tests/test_gpu_group_by_ident.py -m gpu -x -q 2>&1 | tail -3
bash scripts/group_libs.sh glibs3 "base fin512"
echo "--- direct stores"; SLIMM_GROUP_STAGED=0 bash scripts/group_libs.sh glibs3b "base"
echo "--- width 9"; SLIMM_GROUP_WIDTH=9 bash scripts/group_libs.sh glibs3c "base"
echo "--- config3"; CONFIG=config3 bash scripts/group_libs.sh glibs3d "base"
echo "--- config3 direct"; SLIMM_GROUP_STAGED=0 CONFIG=config3 bash scripts/group_libs.sh glibs3e "base"
