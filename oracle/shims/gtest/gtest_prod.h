// Oracle shim (test infrastructure, NOT product code).
// Stands in for googletest's gtest_prod.h so that the reference's
// yadcc/scheduler/task_dispatcher.h (line 26, 277-278) compiles unmodified.
#ifndef ORACLE_SHIM_GTEST_PROD_H_
#define ORACLE_SHIM_GTEST_PROD_H_
#define FRIEND_TEST(suite, name) friend class suite##_##name##_Test
#endif
