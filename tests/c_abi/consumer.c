/* A plain-C consumer of include/mi355_vllm.h, the way cgo / bindgen / a C host would bind it: compiled by gcc as C11
 * (-pedantic -Werror), linked against libmi355vllm.so, and calling host-only entry points (nothing here needs a GPU).
 * Prints "key value" lines that tests/test_cpu_c_abi.py compares with the ctypes view of the same library. */
#include <stdio.h>
#include <string.h>

#include "mi355_vllm.h"

int main(void) {
    /* struct layouts as THIS compiler sees the header vs what the library was built with */
    printf("sizeof_qmm_desc %d %d\n", (int)sizeof(mi355_qmm_desc), (int)mi355_abi_struct_size(0));
    printf("sizeof_llama_config %d %d\n", (int)sizeof(mi355_llama_config), (int)mi355_abi_struct_size(1));
    printf("sizeof_dense_config %d %d\n", (int)sizeof(mi355_dense_config), (int)mi355_abi_struct_size(2));
    printf("sizeof_rope_scaling %d %d\n", (int)sizeof(mi355_rope_scaling), (int)mi355_abi_struct_size(3));
    printf("unknown_struct %d\n", (int)mi355_abi_struct_size(99));

    /* GGUF tile repack sizes: Q4_K 2304 B, Q6_K 3360 B per 16x256 tile */
    printf("repacked_q4k %lld\n", (long long)mi355_qweight_repacked_size(MI355_GGML_Q4_K, 32, 512));
    printf("repacked_q6k %lld\n", (long long)mi355_qweight_repacked_size(MI355_GGML_Q6_K, 32, 512));
    printf("repacked_bad %lld\n", (long long)mi355_qweight_repacked_size(MI355_GGML_Q4_K, 32, 100));

    /* RoPE tables (host code): llama3 scaling keeps the table length, row 0 is cos 1 / sin 0 */
    {
        mi355_rope_scaling sc;
        float cosv[4 * 4], sinv[4 * 4];
        memset(&sc, 0, sizeof(sc));
        sc.type = MI355_ROPE_LLAMA3;
        sc.factor = 8.0; sc.low_freq_factor = 1.0; sc.high_freq_factor = 4.0; sc.original_max_position_embeddings = 8192.0;
        printf("rope_len %d\n", (int)mi355_rope_table_len(&sc, 4, 131072));
        printf("rope_rc %d\n", mi355_rope_tables(cosv, sinv, 8, 4, 500000.0, &sc, 4, 131072));
        printf("rope_row0 %.1f %.1f\n", (double)cosv[0], (double)sinv[0]);
    }
    /* a missing file is reported, not crashed on */
    printf("gguf_open_missing %d\n", mi355_gguf_open("/nonexistent/file.gguf") == NULL);
    return 0;
}
