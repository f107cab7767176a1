/*
 * nsk_threads.h — how many host threads the OpenMP loops of the two libraries should use.
 *
 * A container often sees every CPU of the machine (256 on the MI355X boxes) while its cgroup grants far fewer
 * (16): an OpenMP team sized after the former makes a 1 ms loop take seconds.  Unless the user set
 * OMP_NUM_THREADS, both libraries cap their teams at the cgroup CPU quota (cgroup v2 cpu.max, v1 cfs quota),
 * the affinity mask and 64.
 */
#ifndef NSK_THREADS_H
#define NSK_THREADS_H

#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int nsk_cpu_budget(void) {
  int budget = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) budget = CPU_COUNT(&set);
  if (budget <= 0) budget = 1;
  double quota = -1.0, period = 100000.0;
  FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[64] = {0};
    if (fscanf(f, "%63s %lf", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atof(q);
    fclose(f);
  } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {
    if (fscanf(f, "%lf", &quota) != 1) quota = -1.0;
    fclose(f);
    if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) {
      if (fscanf(f, "%lf", &period) != 1) period = 100000.0;
      fclose(f);
    }
  }
  if (quota > 0.0 && period > 0.0) {
    const int q = (int)((quota + period - 1.0) / period);
    if (q >= 1 && q < budget) budget = q;
  }
  return budget > 64 ? 64 : budget;
}

#endif
