// cmf_newton.hip.h -- Newton solver entry points (included by cmf_api.hip)
extern "C" int cmf_newton_step(cmf_ctx *c, double, double, double, int, int, int, int, double, double,
                               const int32_t *, const int32_t *, const int32_t *, const int32_t *) {
    NEED_PROBLEM(c);
    return fail(CMF_EUNSUPPORTED, "newton step not built yet");
}
extern "C" int cmf_newton_uz_update(cmf_ctx *c, double, double, double, int, int, double) {
    NEED_PROBLEM(c);
    return fail(CMF_EUNSUPPORTED, "newton step not built yet");
}
extern "C" int cmf_newton_v_partials(cmf_ctx *c, double, float *) {
    NEED_PROBLEM(c);
    return fail(CMF_EUNSUPPORTED, "newton step not built yet");
}
extern "C" int cmf_newton_v_apply(cmf_ctx *c, const float *, double, double, int, double) {
    NEED_PROBLEM(c);
    return fail(CMF_EUNSUPPORTED, "newton step not built yet");
}
extern "C" int cmf_safe_invert_batch(cmf_ctx *c, const double *, double *, int, int, double) {
    if (!c) return fail(CMF_EINVAL, "null context");
    return fail(CMF_EUNSUPPORTED, "eigen solver not built yet");
}
