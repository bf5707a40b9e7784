// deposit circuit (src/deposit/circuit/*.tcc) — placeholder until the Merkle gadgets land
#include <stdexcept>
#include "blockmaze_circuits.hpp"
namespace zk {
std::unique_ptr<Circuit> make_deposit_circuit(bool, size_t) { throw std::runtime_error("deposit circuit: not implemented yet"); }
void assign_deposit(Circuit &, const DepositInputs &) { throw std::runtime_error("deposit circuit: not implemented yet"); }
}
