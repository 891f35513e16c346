// stand-in for <sqlite3.h>: every call throws (loadFromColmapDB is never run by the oracle glue; global-lvba_amd/dataset.py
// reads COLMAP databases with Python's sqlite3).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "lvba_unavailable.h"
typedef struct sqlite3 sqlite3;
typedef struct sqlite3_stmt sqlite3_stmt;
typedef long long sqlite3_int64;
#define SQLITE_OK 0
#define SQLITE_ROW 100
#define SQLITE_DONE 101
#define SQLITE_OPEN_READONLY 1
inline int sqlite3_open(const char *, sqlite3 **) { lvba_unavailable("sqlite3_open"); }
inline int sqlite3_open_v2(const char *, sqlite3 **, int, const char *) { lvba_unavailable("sqlite3_open_v2"); }
inline int sqlite3_close(sqlite3 *) { lvba_unavailable("sqlite3_close"); }
inline const char *sqlite3_errmsg(sqlite3 *) { lvba_unavailable("sqlite3_errmsg"); }
inline int sqlite3_prepare_v2(sqlite3 *, const char *, int, sqlite3_stmt **, const char **) { lvba_unavailable("sqlite3_prepare_v2"); }
inline int sqlite3_step(sqlite3_stmt *) { lvba_unavailable("sqlite3_step"); }
inline int sqlite3_reset(sqlite3_stmt *) { lvba_unavailable("sqlite3_reset"); }
inline int sqlite3_finalize(sqlite3_stmt *) { lvba_unavailable("sqlite3_finalize"); }
inline int sqlite3_bind_int(sqlite3_stmt *, int, int) { lvba_unavailable("sqlite3_bind_int"); }
inline int sqlite3_bind_int64(sqlite3_stmt *, int, sqlite3_int64) { lvba_unavailable("sqlite3_bind_int64"); }
inline int sqlite3_column_int(sqlite3_stmt *, int) { lvba_unavailable("sqlite3_column_int"); }
inline sqlite3_int64 sqlite3_column_int64(sqlite3_stmt *, int) { lvba_unavailable("sqlite3_column_int64"); }
inline const void *sqlite3_column_blob(sqlite3_stmt *, int) { lvba_unavailable("sqlite3_column_blob"); }
inline int sqlite3_column_bytes(sqlite3_stmt *, int) { lvba_unavailable("sqlite3_column_bytes"); }
inline const unsigned char *sqlite3_column_text(sqlite3_stmt *, int) { lvba_unavailable("sqlite3_column_text"); }
