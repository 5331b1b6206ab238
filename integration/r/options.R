# options.R -- run-time knobs of the HIP backend (read by src/gpirt_shim.c through GetOption1()).
# The package's own R code needs no change: R/gpirtMCMC.R and the one-line `.gpirtMCMC` wrapper in
# R/RcppExports.R keep calling `.Call("_gpirt_gpirtMCMC", ...)` with the same seven arguments.
#
#   options(gpirt.hip.rng = "reference")        # default: replay R's Mersenne-Twister stream draw for draw
#   options(gpirt.hip.rng = "item")             # batched counter-based RNG keyed by (seed, iteration, stage, item)
#   options(gpirt.hip.theta_stabilise = TRUE)   # subtract the row maximum before exp() in draw_theta
#   options(gpirt.hip.fstar_fused = TRUE)       # predictive mean as (L^-1 k*)^T (L^-1 f)
#   options(gpirt.hip.kstar_rank = 64L)         # with fstar_fused: K(theta, theta*) through its exact rank-64
#                                               # Chebyshev factorisation (2 x 64 solves instead of 1001 + m)
.onLoad <- function(libname, pkgname) {
    op <- list(gpirt.hip.rng = "reference", gpirt.hip.theta_stabilise = FALSE, gpirt.hip.fstar_fused = FALSE,
               gpirt.hip.kstar_rank = 0L)
    toset <- !(names(op) %in% names(options()))
    if (any(toset)) options(op[toset])
    invisible()
}
