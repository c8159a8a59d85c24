// direct.hip -- direct KKT back-end (rows K2-K4): placeholder until the LDL' path lands.
#include "engine.hpp"
namespace oq {
std::unique_ptr<Linsys> make_direct(Engine &e, int *err) { *err = -1; return nullptr; }
int polish_run(Engine &e) { return 0; }
}  // namespace oq
