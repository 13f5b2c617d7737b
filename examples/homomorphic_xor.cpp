// The reference crate's examples/homomorphic_xor.rs on the MI355X path:
// XOR of an encrypted bit with a constant, then decryption.
//
// The crate's example runs with its `bootstrapping` feature off; here the gate
// IS bootstrapped on the GPU.  With keys generated the way the crate generates
// them the bootstrap degenerates (its external product is identically zero) and
// the result is the trivial sample +-1/8 whose sign follows the input's body,
// so the decrypted bit is printed, not asserted: it is checked against the CPU
// restatement of the crate in tests/host_api_test.cpp.
#include <cstdio>

#include "../tfhe_amd/host/tfhe.hpp"

int main() {
  using namespace tfhe;
  const Parameters params = Parameters::with(SecurityLevel::Bit80);
  auto [secret_key, cloud_key] = generate_keys(params);
  const LweSample encrypted = encrypt(false, secret_key);
  const LweSample one = constant(true, cloud_key);
  const LweSample result = xor_(encrypted, one, cloud_key);
  std::printf("xor(enc(false), constant(true)) on the GPU: b = %d, variance = %g, decrypts to %s\n",
              result.b, result.current_variance, decrypt(result, secret_key) ? "true" : "false");
  const LweSample plain = not_(encrypted, cloud_key);
  std::printf("not(enc(false)) decrypts to %s\n", decrypt(plain, secret_key) ? "true" : "false");
  return decrypt(plain, secret_key) ? 0 : 1;
}
