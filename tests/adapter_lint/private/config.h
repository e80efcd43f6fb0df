/* LINT ONLY: the values cmake writes into RDycore's include/private/config.h (CMakeLists.txt:11-22) */
#ifndef CONFIG_H
#define CONFIG_H
#define MAX_NAME_LEN 128
#define MAX_NUM_FIELDS 10
#define MAX_NUM_FIELD_COMPONENTS 10
#define MAX_NUM_SEDIMENT_CLASSES 5
#define MAX_NUM_TRACERS 7
#define MATERIAL_PROPERTY_MANNINGS 0
#define NUM_MATERIAL_PROPERTIES 1
#define PETSC_ID_TYPE "int32"
#endif
