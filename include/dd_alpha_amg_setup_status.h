/*
 * dd_alpha_amg_setup_status.h -- setup ageing counters, ABI compatible with the reference's
 * src/dd_alpha_amg_setup_status.h:25-28.
 */
#ifndef DDaplhaAMG_SETUP_STATUS_H
#define DDaplhaAMG_SETUP_STATUS_H

typedef struct dd_alpha_amg_setup_status {
  int gauge_updates_since_last_setup;
  int gauge_updates_since_last_setup_update;
} dd_alpha_amg_setup_status;

#endif
