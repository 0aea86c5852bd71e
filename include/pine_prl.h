/* pine_prl.h -- C ABI of the minimal PRL front-end (libpine_prl.so).
 *
 * Replaces, for cbox-class scripts, the reference's `jit_interpret(Context&, source)`
 * (src/pine/core/jit.cpp:2219-2355: parse -> LLVM IR -> MCJIT -> run) and the CLI's
 * `interpret_file` (src/pine/core/fileio.cpp:575-579, src/cli/pine.cpp:27): the script is parsed
 * with the reference's grammar (jit.cpp:1467-2217) and executed by a tree-walking interpreter whose
 * function table mirrors the names, overloads and one-step implicit conversions registered in
 * setup_program_context() (src/pine/core/program_context.cpp:23-125) for the PathIntegrator path;
 * `PathIntegrator(...).render(scene)` runs on the MI355X through libpine_gpu.so.
 * Statements, loops, `fn` definitions with `return`; no `class`, no lambdas.
 * No LLVM, no CPU rendering fallback.
 */
#ifndef PINE_PRL_H
#define PINE_PRL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PINE_PRL_DRY_RUN 1 /* do not touch the GPU: `render` records the scene description and its
                              arguments in the output log, `save` records the file name           */
#define PINE_PRL_ECHO    2 /* copy print/println output to stdout as well                         */

/* Runs `source` (a whole .pine script).  Everything the script prints, and in dry-run mode the
 * render/save records, is appended to an internal log that pine_prl_output() returns (valid until
 * the next call on this thread).  device: HIP device ordinal for render().
 * Returns 0, or a negative value with the message (line:column + text) in pine_prl_last_error(). */
int pine_prl_interpret(const char* source, int flags, int device);
const char* pine_prl_output(void);
const char* pine_prl_last_error(void);

/* The film (W*H float4, row 0 first) of the last `render` executed by pine_prl_interpret on this thread:
 * lets a host read the float film without going through the script's 8-bit `save`.  Returns the
 * number of floats (0 if nothing was rendered); the pointer stays valid until the next interpret call. */
int64_t pine_prl_last_film(const float** data, int* width, int* height);

/* Evaluates one PRL expression and writes "<type> <value>" (floats as C99 hex floats) to `out`;
 * a test hook for literal / operator / overload semantics.  Returns the length needed. */
int64_t pine_prl_eval(const char* expression, char* out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* PINE_PRL_H */
