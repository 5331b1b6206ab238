# gpirtMCMC.R -- the R wrapper stays EXACTLY the reference's (R/gpirtMCMC.R:85-105): same formals,
# same lazy defaults (ncol(data) is evaluated after `data` was re-assigned, quirk Q8), theta_init
# drawn in R before the call.  Only the native routine behind .gpirtMCMC changes (gpirt_shim.c).
gpirtMCMC <- function(data, sample_iterations, burn_iterations,
                      vote_codes = list(yea = 1:3, nay = 4:6,
                                        missing = c(0, 7:9, NA)),
                      beta_prior_means = matrix(0, nrow = 2, ncol = ncol(data)),
                      beta_prior_sds = matrix(3, nrow = 2, ncol = ncol(data)),
                      beta_proposal_sds = matrix(0.1, nrow = 2, ncol = ncol(data)),
                      theta_init = NULL) {
    data <- as.response_matrix(data, vote_codes)
    if ( is.null(theta_init) ) {
        theta_init  <- rnorm(nrow(data))
    }
    storage.mode(data) <- "double"
    .Call(`_gpirt_gpirtMCMC`, data, as.double(theta_init),
          as.integer(sample_iterations), as.integer(burn_iterations),
          beta_prior_means, beta_prior_sds, beta_proposal_sds)
}
# Extra knobs never become new required arguments:
#   options(gpirt.hip.rng = "item")             # batched counter-based RNG (default: replay R's stream)
#   options(gpirt.hip.theta_stabilise = TRUE)   # row-max shift in draw_theta (defined where the reference underflows)
#   options(gpirt.hip.fstar_fused = TRUE)       # mean = (L^-1 k*)^T (L^-1 f)
