for wl in C4b6 C4; do
  for v in nolik_occ6 dyn_occ7 dyn_occ6 dyn_occ5 nolik_p40_occ5; do
    MPC_AB_LIB=build_ab/$v.so python tools/step_probe.py $wl --steps 20 2>&1 | grep -v amdgpu
  done
done
