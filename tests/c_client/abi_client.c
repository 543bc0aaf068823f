/* abi_client.c -- a plain C99 program that drives libcppflow_hip.so through include/cppflow_hip.h and the HIP runtime's C
 * API only: no Python, no torch, no C++.  It is what a non-Python caller of the boundary looks like, and the test
 * tests/test_gpu_c_client.py checks that it produces exactly what the Python mirror produces for the same inputs.
 *
 *   abi_client <in.bin> <out.bin>
 *
 * in.bin  = int32 S, W, K, n_obs | cppf_robot_desc | float cuboids[n_obs*6] | float Rt[n_obs*12] | float jl_lo[16], jl_hi[16]
 *           | float x0[S*W*d] | float target[W*7]
 * out.bin = float x[S*W*d] | float ext_cost[n] | float pos_err[n] | float rot_err[n] | u8 self[n] | u8 env[n] | u8 jlim[n]
 *           | float summary[S*8] | float fk[n*7]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cppflow_hip.h"

#define HIP_OK(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return 2;                                                                       \
        }                                                                                   \
    } while (0)

#define CPPF_OK_OR_DIE(call)                                                                 \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != 0) {                                                                      \
            fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, cppf_last_error()); \
            return 3;                                                                        \
        }                                                                                    \
    } while (0)

static int read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 1;
    }
    if (cppf_abi_version() != CPPF_ABI_VERSION) {
        fprintf(stderr, "header / library ABI mismatch: %d vs %d\n", CPPF_ABI_VERSION, cppf_abi_version());
        return 1;
    }
    FILE* in = fopen(argv[1], "rb");
    if (!in) return 1;
    int32_t hdr[4];
    cppf_robot_desc desc;
    float cuboids[CPPF_MAX_OBSTACLES * 6], Rt[CPPF_MAX_OBSTACLES * 12], jl_lo[CPPF_MAX_DOF], jl_hi[CPPF_MAX_DOF];
    if (read_exact(in, hdr, sizeof hdr) || read_exact(in, &desc, sizeof desc)) return 1;
    const int S = hdr[0], W = hdr[1], K = hdr[2], n_obs = hdr[3], d = desc.ndof;
    if (n_obs < 0 || n_obs > CPPF_MAX_OBSTACLES || S < 1 || W < 1 || K < 1) return 1;
    if (read_exact(in, cuboids, sizeof(float) * 6 * n_obs) || read_exact(in, Rt, sizeof(float) * 12 * n_obs) ||
        read_exact(in, jl_lo, sizeof jl_lo) || read_exact(in, jl_hi, sizeof jl_hi))
        return 1;
    const size_t n = (size_t)S * W;
    float* h_x = (float*)malloc(sizeof(float) * n * d);
    float* h_t = (float*)malloc(sizeof(float) * W * 7);
    if (!h_x || !h_t || read_exact(in, h_x, sizeof(float) * n * d) || read_exact(in, h_t, sizeof(float) * W * 7)) return 1;
    fclose(in);

    cppf_robot* robot = NULL;
    CPPF_OK_OR_DIE(cppf_robot_create(&desc, 0, &robot));
    CPPF_OK_OR_DIE(cppf_set_obstacles(robot, n_obs, cuboids, Rt));
    CPPF_OK_OR_DIE(cppf_set_joint_limit_padding(robot, jl_lo, jl_hi));

    hipStream_t stream;
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipStreamCreate(&stream));
    float *x_in, *x_out, *target, *cost, *pe, *re, *summary, *fk;
    uint8_t *m_self, *m_env, *m_jl;
    HIP_OK(hipMalloc((void**)&x_in, sizeof(float) * n * d));
    HIP_OK(hipMalloc((void**)&x_out, sizeof(float) * n * d));
    HIP_OK(hipMalloc((void**)&target, sizeof(float) * W * 7));
    HIP_OK(hipMalloc((void**)&cost, sizeof(float) * n));
    HIP_OK(hipMalloc((void**)&pe, sizeof(float) * n));
    HIP_OK(hipMalloc((void**)&re, sizeof(float) * n));
    HIP_OK(hipMalloc((void**)&summary, sizeof(float) * S * 8));
    HIP_OK(hipMalloc((void**)&fk, sizeof(float) * n * 7));
    HIP_OK(hipMalloc((void**)&m_self, n));
    HIP_OK(hipMalloc((void**)&m_env, n));
    HIP_OK(hipMalloc((void**)&m_jl, n));
    HIP_OK(hipMemcpyAsync(x_in, h_x, sizeof(float) * n * d, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(target, h_t, sizeof(float) * W * 7, hipMemcpyHostToDevice, stream));

    cppf_lm_params prm;
    memset(&prm, 0, sizeof prm); /* early-out off, shape = CPPF_SHAPE_AUTO */
    prm.lm_lambda = 1e-6f, prm.alpha_position = 3.5f, prm.alpha_rotation = 0.35f; /* ALT_LOSS_V2_1_POSE */
    prm.n_steps = K, prm.clamp = 1;
    cppf_lm_outputs out;
    memset(&out, 0, sizeof out);
    out.x_out = x_out, out.pos_err_m = pe, out.rot_err_rad = re, out.self_mask = m_self, out.env_mask = m_env;
    out.jlim_mask = m_jl, out.ext_cost = cost, out.seed_summary = summary;
    CPPF_OK_OR_DIE(cppf_lm_pose_steps(robot, x_in, target, S, W, &prm, &out, stream));
    CPPF_OK_OR_DIE(cppf_forward_kinematics(robot, x_out, (int)n, fk, stream));

    /* contract violations come back as CPPF_ERR_INVALID (-1) with a message, before any launch */
    prm.n_steps = 0;
    if (cppf_lm_pose_steps(robot, x_in, target, S, W, &prm, &out, stream) != -1 || strlen(cppf_last_error()) == 0) {
        fprintf(stderr, "n_steps = 0 was not rejected\n");
        return 4;
    }

    const size_t out_bytes = sizeof(float) * (n * d + 3 * n + (size_t)S * 8 + n * 7) + 3 * n;
    unsigned char* h_out = (unsigned char*)malloc(out_bytes);
    unsigned char* p = h_out;
    if (!h_out) return 1;
#define PULL(src, bytes)                                                            \
    HIP_OK(hipMemcpyAsync(p, (src), (bytes), hipMemcpyDeviceToHost, stream)); \
    p += (bytes)
    PULL(x_out, sizeof(float) * n * d);
    PULL(cost, sizeof(float) * n);
    PULL(pe, sizeof(float) * n);
    PULL(re, sizeof(float) * n);
    PULL(m_self, n);
    PULL(m_env, n);
    PULL(m_jl, n);
    PULL(summary, sizeof(float) * S * 8);
    PULL(fk, sizeof(float) * n * 7);
    HIP_OK(hipStreamSynchronize(stream));
    FILE* fo = fopen(argv[2], "wb");
    if (!fo || fwrite(h_out, 1, out_bytes, fo) != out_bytes) return 1;
    fclose(fo);

    /* ---- seed sharding through the C ABI: R communicators in this one process (cppf_comm_init_all), R = CPPF_CLIENT_RANKS capped
     * by the visible devices (1 on a one-GPU box).  Rank r holds the external cost of seeds [r S/R, (r+1) S/R) on device r; ONE
     * all-gather per rank inside a group; every rank must end up with all S*W costs in seed order. ---- */
    {
        int ndev = 0, R = 1;
        HIP_OK(hipGetDeviceCount(&ndev));
        const char* env = getenv("CPPF_CLIENT_RANKS");
        if (env) R = atoi(env);
        if (R > ndev) R = ndev;
        while (R > 1 && S % R != 0) --R;
        int devs[8];
        cppf_comm* comms[8];
        float* d_send[8];
        float* d_recv[8];
        hipStream_t st[8];
        if (R > 8) R = 8;
        for (int r = 0; r < R; ++r) devs[r] = r;
        CPPF_OK_OR_DIE(cppf_comm_init_all(R, devs, comms));
        const size_t per = n / R;
        const float* h_cost = (const float*)(h_out + sizeof(float) * n * d);
        for (int r = 0; r < R; ++r) {
            HIP_OK(hipSetDevice(r));
            HIP_OK(hipStreamCreate(&st[r]));
            HIP_OK(hipMalloc((void**)&d_send[r], sizeof(float) * per));
            HIP_OK(hipMalloc((void**)&d_recv[r], sizeof(float) * n));
            HIP_OK(hipMemcpyAsync(d_send[r], h_cost + r * per, sizeof(float) * per, hipMemcpyHostToDevice, st[r]));
            if (cppf_comm_rank(comms[r]) != r || cppf_comm_world(comms[r]) != R) return 5;
        }
        CPPF_OK_OR_DIE(cppf_comm_group_begin());
        for (int r = 0; r < R; ++r) CPPF_OK_OR_DIE(cppf_allgather_bytes(comms[r], d_send[r], d_recv[r], sizeof(float) * per, st[r]));
        CPPF_OK_OR_DIE(cppf_comm_group_end());
        float* h_back = (float*)malloc(sizeof(float) * n);
        for (int r = 0; r < R; ++r) {
            HIP_OK(hipSetDevice(r));
            HIP_OK(hipMemcpyAsync(h_back, d_recv[r], sizeof(float) * n, hipMemcpyDeviceToHost, st[r]));
            HIP_OK(hipStreamSynchronize(st[r]));
            if (memcmp(h_back, h_cost, sizeof(float) * n) != 0) {
                fprintf(stderr, "rank %d: gathered costs differ\n", r);
                return 6;
            }
        }
        for (int r = 0; r < R; ++r) {
            HIP_OK(hipSetDevice(r));
            cppf_comm_destroy(comms[r]);
            hipFree(d_send[r]), hipFree(d_recv[r]);
            hipStreamDestroy(st[r]);
        }
        free(h_back);
        HIP_OK(hipSetDevice(0));
        printf("allgather ok on %d rank(s)\n", R);
    }

    /* ---- lifetimes (cppflow_hip.h, "Ownership"): a batch keeps its robot allocated.  Destroying the robot FIRST is allowed: the
     * batch's launches then fail with CPPF_ERR_INVALID (no launch, no fault), and destroying the batch releases both. ---- */
    {
        cppf_lm_batch_item item;
        memset(&item, 0, sizeof item);
        item.x_in = x_in, item.target = target, item.S = S, item.W = W;
        item.out.x_out = x_out, item.out.pos_err_m = pe, item.out.rot_err_rad = re;
        prm.n_steps = K;
        cppf_lm_batch* batch = NULL;
        CPPF_OK_OR_DIE(cppf_lm_batch_create(robot, 1, &item, &prm, &batch));
        CPPF_OK_OR_DIE(cppf_lm_batch_launch(batch, stream)); /* alive: a normal launch, same results as above */
        HIP_OK(hipStreamSynchronize(stream));
        float* h_chk = (float*)malloc(sizeof(float) * n * d);
        HIP_OK(hipMemcpy(h_chk, x_out, sizeof(float) * n * d, hipMemcpyDeviceToHost));
        if (memcmp(h_chk, h_out, sizeof(float) * n * d) != 0) {
            fprintf(stderr, "batched launch differs from cppf_lm_pose_steps\n");
            return 7;
        }
        free(h_chk);
        cppf_robot_destroy(robot); /* robot first ... */
        if (cppf_lm_batch_launch(batch, stream) != CPPF_ERR_INVALID || strlen(cppf_last_error()) == 0) {
            fprintf(stderr, "a launch on a destroyed robot was not rejected\n");
            return 8;
        }
        if (cppf_forward_kinematics(robot, x_out, (int)n, fk, stream) != CPPF_ERR_INVALID) { /* (still allocated: the batch holds it) */
            fprintf(stderr, "an entry point on a destroyed robot was not rejected\n");
            return 8;
        }
        cppf_lm_batch_destroy(batch); /* ... then the batch: releases the handle */
        printf("lifetime ok\n");
    }
    hipFree(x_in), hipFree(x_out), hipFree(target), hipFree(cost), hipFree(pe), hipFree(re), hipFree(summary), hipFree(fk);
    hipFree(m_self), hipFree(m_env), hipFree(m_jl);
    hipStreamDestroy(stream);
    free(h_x), free(h_t), free(h_out);
    printf("ok %zu rows\n", n);
    return 0;
}
