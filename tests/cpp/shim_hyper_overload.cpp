// Compile-only check (tests/test_host_logic.py): a non-const gs_hyper lvalue selects Trainer::train(const gs_hyper&, bool)
// -> gs_step_stats, not the Project template (round-2 advisor finding); the Project form still binds to Project.
#include <type_traits>

#include "gsplat_shim.hpp"

using namespace gsplat_shim;

void overloads(Trainer& t, Project& project) {
    gs_hyper hy{};
    gs_hyper_defaults(&hy);
    gs_step_stats st = t.train(hy, false);   // non-const lvalue
    const gs_hyper chy = hy;
    st = t.train(chy, true);
    (void)st;
    static_assert(std::is_same<decltype(t.train(hy, false)), gs_step_stats>::value, "gs_hyper& must take the statistics-returning overload");
    static_assert(std::is_same<decltype(t.train(project, false)), void>::value, "Project takes the reference-shaped overload");
    t.train(project, false);
}
