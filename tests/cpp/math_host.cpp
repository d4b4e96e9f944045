// Host build of geosss_amd/csrc/gsss_math.h for tests/test_math.py (plain g++, no HIP).
#include "../../geosss_amd/csrc/gsss_math.h"
extern "C" {
void t_sincos_small(const double *x, long n, double *s, double *c) { for (long i = 0; i < n; ++i) gsss::fm::sincos_small(x[i], s[i], c[i]); }
void t_sincos_2pi(const double *x, long n, double *s, double *c) { for (long i = 0; i < n; ++i) gsss::fm::sincos_2pi(x[i], s[i], c[i]); }
void t_exp(const double *x, long n, double *y) { for (long i = 0; i < n; ++i) y[i] = gsss::fm::exp_fast(x[i]); }
void t_exp_bounded(const double *x, long n, double *y) { for (long i = 0; i < n; ++i) y[i] = gsss::fm::exp_bounded(x[i]); }
void t_log(const double *x, long n, double *y) { for (long i = 0; i < n; ++i) y[i] = gsss::fm::log_fast(x[i]); }
}
// table-driven variants (tables built exactly as the kernels build them)
static gsss::fm::Tables host_tables()
{
    static double buf[gsss::fm::kTableDoubles];
    static bool done = false;
    if (!done) {
        for (int i = 0; i < 64 + gsss::fm::kLogTableN; ++i) gsss::fm::table_entry(buf, i);
        done = true;
    }
    return gsss::fm::Tables{buf, buf + 128};
}
extern "C" {
void t_sincos_tab(const double *x, long n, double *s, double *c) { auto t = host_tables(); for (long i = 0; i < n; ++i) gsss::fm::sincos_tab(x[i], t, s[i], c[i]); }
void t_sincos_word_tab(const unsigned *w, long n, double *s, double *c) { auto t = host_tables(); for (long i = 0; i < n; ++i) gsss::fm::sincos_word_tab(w[i], t, s[i], c[i]); }
void t_log_word_tab(const unsigned *w, long n, double *y) { auto t = host_tables(); for (long i = 0; i < n; ++i) y[i] = gsss::fm::log_word_tab(w[i], t); }
}

extern "C" void t_box_muller_f32(const unsigned *wr, const unsigned *wa, long n, double *z0, double *z1)
{
    for (long i = 0; i < n; ++i) {
        float a, b;
        gsss::fm::box_muller_f32(wr[i], wa[i], a, b);
        z0[i] = a;
        z1[i] = b;
    }
}
