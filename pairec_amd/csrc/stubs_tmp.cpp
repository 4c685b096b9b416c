// temporary: entry points not implemented yet return PG_ERR_UNSUPPORTED
#include "common.hpp"
#define STUB(name, ...) int name(__VA_ARGS__) { pg::set_error(#name ": not implemented yet"); return PG_ERR_UNSUPPORTED; }
extern "C" {
STUB(pg_expr_compile, const char*, pg_expr**)
STUB(pg_expr_free, pg_expr*)
STUB(pg_expr_num_vars, const pg_expr*)
const char* pg_expr_var_name(const pg_expr*, int) { return ""; }
STUB(pg_expr_eval, pg_ctx*, const pg_expr*, const double*, uint32_t, double*)
STUB(pg_expr_eval_dev, pg_ctx*, const pg_expr*, const double*, uint32_t, double*)
STUB(pg_sort_scores, pg_ctx*, const double*, const uint32_t*, uint32_t, int, uint32_t*)
STUB(pg_sort_scores_dev, pg_ctx*, const double*, const uint32_t*, uint32_t, uint32_t, int, uint32_t*)
STUB(pg_dpp, pg_ctx*, const pg_table*, const uint32_t*, const double*, uint32_t, double, uint32_t, uint32_t, int, uint32_t*, uint32_t*)
}
