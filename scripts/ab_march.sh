# A/B of the marching inverse FFT (RBC_IFFT_MARCH = 0: k3_ifft_pair + k3_correct_w; 2: k3_ifft_march) at configs[4], interleaved on one box
for i in 1 2 3; do
  for m in 0 2; do for p in f64 f32; do echo -n "march=$m "; RBC_IFFT_MARCH=$m python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done; done
done
