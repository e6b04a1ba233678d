#include <atomic>
#include <cstdlib>
#include "common.h"
#include "lc2is_hip.h"
extern "C" const char* lc2is_version(void) { return "lc2is_hip 1 gfx950"; }

// CU budget of the tile planners (common.h: lc2is_ncu).  Process-wide; the data-parallel reducer sets it while RCCL's channels hold CUs.
static int cu_budget_env() {   // LC2IS_CU_BUDGET=<n>: the initial budget (A/B: what reserving CUs costs a single-GPU step)
  const char* e = getenv("LC2IS_CU_BUDGET");
  const int v = e ? atoi(e) : 0;
  return v > 0 && v <= 256 ? v : 0;
}
static std::atomic<int> g_cu_budget{cu_budget_env()};
extern "C" int lc2is_set_cu_budget(int ncu) {
  if (ncu < 0 || ncu > 256) return LC2IS_ERR_SHAPE;
  g_cu_budget.store(ncu);
  return LC2IS_OK;
}
extern "C" int lc2is_get_cu_budget(void) { return g_cu_budget.load(); }
