set -e
for i in 1 2; do
echo "-- HEAD"; AP_LIB_PATH=build/alt/libaudioprims_head.so python tools/time_op.py istftrows 2>&1 | grep -v amdgpu.ids; AP_LIB_PATH=build/alt/libaudioprims_head.so python tools/time_op.py istftd 2>&1 | grep -v amdgpu.ids
echo "-- new"; python tools/time_op.py istftrows 2>&1 | grep -v amdgpu.ids; python tools/time_op.py istftd 2>&1 | grep -v amdgpu.ids
done
AP_LIB_PATH=build/alt/libaudioprims_head.so python tools/time_gl.py 2>&1 | grep -v amdgpu.ids
python tools/time_gl.py 2>&1 | grep -v amdgpu.ids
