for sd in 300005 304767 306981; do
 for fw in 1 0; do
  echo "== seed $sd fused_walk=$fw: $(ODDIO_SOAK_ONLY=test_random_operations_bit_exact ODDIO_FUZZ_MODE=fast ODDIO_HIP_PAIR_MIN_GROUPS=1 ODDIO_HIP_FUSED_WALK=$fw python tests/soak_fuzz.py $sd 1 2>&1 | grep -v amdgpu | tail -3 | cut -c1-600 | tr '\n' ' ')"
 done
done
