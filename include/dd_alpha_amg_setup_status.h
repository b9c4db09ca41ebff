/*
 * Interface declarations (struct fields, prototypes, include guards) follow the DDalphaAMG solver library:
 * Copyright (C) 2016, Matthias Rottmann, Artur Strebel, Simon Heybrock, Simone Bacchio, Bjoern Leder, Issaku Kanamori.
 *
 * The DDalphaAMG solver library is free software: you can redistribute it and/or modify
 * it under the terms of the GNU General Public License as published by
 * the Free Software Foundation, either version 3 of the License, or
 * (at your option) any later version.
 *
 * The DDalphaAMG solver library is distributed in the hope that it will be useful,
 * but WITHOUT ANY WARRANTY; without even the implied warranty of
 * MERCHANTABILITY or FITNESS FOR A PARTICULAR PURPOSE.  See the
 * GNU General Public License for more details.
 *
 * You should have received a copy of the GNU General Public License
 * along with the DDalphaAMG solver library. If not, see http://www.gnu.org/licenses/.
 *
 * This header reproduces that interface so that the MI355X implementation in this repository drops in behind it; the
 * implementation itself is new code, distributed under the same licence (see LICENSE at the repository root).
 */
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
