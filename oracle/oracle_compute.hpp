// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into or loaded by the product path.
//
// The `Compute` backend of the test harnesses: seam (1) / seam (2) of include/ltpl_hip.h answered by the oracle's CPU arithmetic
// (oracle_plan_paths / oracle_vel_profile of ltpl_oracle.c) instead of the HIP kernels.
#pragma once

#include "../graphbasedlocaltrajectoryplanner_amd/csrc/planner_host.hpp"

extern "C" {
int oracle_plan_paths(const ltpl_lattice_desc* d, const ltpl_paths_in* in, ltpl_paths_out* out);
int oracle_vel_profile(const ltpl_lattice_desc* d, const ltpl_vel_params* params, int n_jobs, const ltpl_vel_job* jobs,
                       ltpl_vel_result* results);
}

namespace {
struct OracleCompute : ltplp::Compute {
    const ltpl_lattice_desc* d;          // owned by the caller (Python keeps the arrays alive)
    std::string err;
    explicit OracleCompute(const ltpl_lattice_desc* desc) : d(desc) {}
    int plan_paths(const ltpl_paths_in* in, ltpl_paths_out* out) override
    {
        const int rc = oracle_plan_paths(d, in, out);
        if (rc) err = "oracle_plan_paths failed";
        return rc;
    }
    int vel_profile(const ltpl_vel_params* p, int n, const ltpl_vel_job* jobs, ltpl_vel_result* res) override
    {
        const int rc = oracle_vel_profile(d, p, n, jobs, res);
        if (rc) err = "oracle_vel_profile failed";
        return rc;
    }
    const char* last_error() override { return err.c_str(); }
};
}  // namespace
