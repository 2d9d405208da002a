/* oracle CLI entry (test infrastructure): same argv as the reference binary,
 * /root/reference/src/bin/lightdock-rust.rs:77-86. */
#include "ld_oracle.h"
int main(int argc, char **argv) { return orc_cli_main(argc, argv); }
