# options.R -- run-time knobs of the HIP backend (read by src/gpirt_shim.c through GetOption1()).
# The package's own R code needs no change: R/gpirtMCMC.R and the one-line `.gpirtMCMC` wrapper in
# R/RcppExports.R keep calling `.Call("_gpirt_gpirtMCMC", ...)` with the same seven arguments.
#
#   options(gpirt.hip.preset = "reference")     # default = gpirt_default_options(): R's Mersenne-Twister stream replayed
#                                               # draw for draw, draw_theta and draw_fstar as written
#   options(gpirt.hip.preset = "fast")          # = gpirt_fast_options(): item-keyed RNG, stabilised draw_theta, fused +
#                                               # rank-64 draw_fstar (what the library's benchmark is quoted on)
# Single knobs, unset (NULL) by default = whatever the preset says:
#   options(gpirt.hip.rng = "reference")        # replay R's stream / "item": batched counter-based RNG keyed by
#                                               # (seed, iteration, stage, item)
#   options(gpirt.hip.theta_stabilise = TRUE)   # subtract the row maximum before exp() in draw_theta
#   options(gpirt.hip.fstar_fused = TRUE)       # predictive mean as (L^-1 k*)^T (L^-1 f)
#   options(gpirt.hip.kstar_rank = 64L)         # with fstar_fused: K(theta, theta*) through its exact rank-64
#                                               # Chebyshev factorisation (2 x 64 solves instead of 1001 + m)
.onLoad <- function(libname, pkgname) {
    op <- list(gpirt.hip.preset = "reference")
    toset <- !(names(op) %in% names(options()))
    if (any(toset)) options(op[toset])
    invisible()
}
