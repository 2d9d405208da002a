// lightdock-hip: the reference's binary name/argv on the MI355X engine.
#include "lightdock_hip.h"
int main(int argc, char **argv) { return ld_cli_main(argc, argv); }
