# gfx950: v_cndmask_b32 in its VOP2 encoding (e32, mask implicit in vcc) issues five times slower than the VOP3 encoding of the same
# operation (tools/ubench/cndmask_rates.hip: 9.8 ns against 1.9 ns per wavefront instruction and SIMD at 1, 2 and 4 wavefronts per SIMD,
# fed-back or independent destination alike).  Re-encode: same operands, same result, four bytes longer.
# (the _dpp / _sdwa forms are other encodings and stay as they are)
s/^([[:space:]]*)v_cndmask_b32_e32 (.*), vcc([[:space:]]*(;.*)?)$/\1v_cndmask_b32_e64 \2, vcc\3/
