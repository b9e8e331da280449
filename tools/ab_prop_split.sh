for so in ab_*.so; do echo $so; THERMONERF_HIP_LIB=$PWD/$so python tools/small_call_forms.py 65536,80000,160000,259200 2>/dev/null | grep lane_ray | grep "S=48"; done
